"""numpy model of the scanner's matrix-core prefilter (radiosonde_auto_rx_amd/csrc/sonde_scan_pre.hip, k_scan_pre) — TEST INFRASTRUCTURE.

Same quantities, same roundings of the operands (window minus 0.98 dc -> f16, FM low-pass taps -> f16, filtered window -> f16, template -> f16,
products accumulated in float32), so that the CPU suite can check the prefilter's one obligation without a GPU: its upper bound
smax = max_p |c[p]| / sqrt(e[p]) never falls below |mv| of the reference's getCorrDFT (oracle/ora_scan.py, pinned to the compiled reference) by more
than the f16 rounding, i.e. no (window, template) the reference would accept is dropped with the 0.03 margin the engine uses."""
import math

import numpy as np

from oracle import ora_scan

F, H = np.float32, np.float16
MARGIN = 0.03


class PrefilterModel:
    def __init__(self, design):
        self.d = design
        sr = design.sr
        taps = int(4 * sr / 2e3)
        taps += (taps % 2 == 0)
        self.ws = [ora_scan.lowpass_taps(F(bw) / F(sr), taps) for bw in (4e3, 10e3)]
        hmax = max(design.hlen[j] for j in design.active)
        self.match = {j: ora_scan.match_template(ora_scan.TEMPLATES[j][1], hmax, design.spb[j], ora_scan.TEMPLATES[j][2], design.L[j])
                      for j in design.active}

    def run(self, j, stream, pos, opt_dc):
        d = self.d
        K, L = d.K, d.L[j]
        wl = K + L
        idx = pos - (wl - 1) + np.arange(wl)
        xn = np.where(idx >= 0, stream[np.clip(idx, 0, len(stream) - 1)], 0).astype(F)
        dc = F(float(np.sum(xn[K - L:K + L], dtype=np.float32)) / (2.0 * L)) if opt_dc else F(0)
        x0 = (xn - F(0.98) * dc).astype(H).astype(F)
        if d.iq:
            ws = self.ws[ora_scan.TEMPLATES[j][7]]
            xf = np.convolve(x0, ws.astype(H).astype(F))[:wl].astype(F)                 # zero history, like the reference's zero padded array
            taps = len(ws)
            tail = np.array([float(np.sum(ws[i + 1:], dtype=np.float64)) for i in range(taps - 1)], F)
            xf[:taps - 1] -= F(0.98) * dc * tail                                       # the constant reaches the filter's first outputs in full
        else:
            xf = x0
        xfh = xf.astype(H).astype(F)
        P = np.concatenate([[0], np.cumsum((xfh * xfh).astype(F), dtype=np.float32)])
        c = np.correlate(xfh, self.match[j].astype(H).astype(F), mode="valid")[:K + 1].astype(F)     # c[p'] = sum_k match[k] xf[p' + k]
        e = (P[L:L + K + 1] - P[:K + 1]).astype(F)
        score = np.where(e > 0, np.abs(c) / np.sqrt(np.maximum(e, 1e-30)), 0)
        pa = int(np.argmax(np.abs(c)))
        mv = float(c[pa] / math.sqrt(max(float(e[pa]), 1e-30)))
        return dict(smax=float(score.max()), mv=mv, mp=(-4 if pa in (0, K) else L - 1 + pa), dc=float(dc))
