"""numpy model of the scanner's matrix-core prefilter (radiosonde_auto_rx_amd/csrc/sonde_scan_pre.hip, k_scan_pre) — TEST INFRASTRUCTURE.

Same quantities, same roundings of the operands (window minus 0.98 dc -> f16, FM low-pass taps -> f16, filtered window -> f16, template -> f16,
products accumulated in float32), so that the CPU suite can check the prefilter's one obligation without a GPU: its upper bound
smax = max_p |c[p]| / sqrt(e[p]) never falls below |mv| of the reference's getCorrDFT (oracle/ora_scan.py, pinned to the compiled reference) by more
than the f16 rounding, i.e. no (window, template) the reference would accept is dropped by the candidate test the engine uses (below)."""
import math

import numpy as np

from oracle import ora_scan

F, H = np.float32, np.float16
# Round 6: the candidate test carries the derived rounding bound per position instead of a flat 0.03:
#   candidate  <=>  max_p [ (|c[p]| + kappa_j X) / sqrt(e[p]) + beta / e[p] ]  >  thres - MARGIN
# kappa_j = 4.02 u ||ws||_1 sqrt(L_j) (u = 2^-11: the f16 roundings of window, taps and filtered window under the template), X = max |window - 0.98 dc|,
# beta = C_P E_win (the float32 prefix sums: C_P = 32 x 2^-24 covers the kernel's three-level sum; E_win = the window's total energy), and MARGIN = 3 u + L 2^-24 +
# the reference's own distance from exact arithmetic (its drifting twiddles: 3e-4) — what does not scale with the signal.
U16 = 2.0 ** -11
C_P = 32 * 2.0 ** -24
MARGIN = 0.003


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

    def kappa(self, j):
        """4.02 u ||ws||_1 sqrt(L): the share of the bound that scales with (window maximum) / (rms under the template); 0 without the FM low-pass (FM audio input)"""
        d = self.d
        if not d.iq:
            return 0.0
        ws = self.ws[ora_scan.TEMPLATES[j][7]]
        return 4.02 * U16 * float(np.sum(np.abs(ws.astype(np.float64)))) * math.sqrt(d.L[j])

    def ideal(self, j, stream, pos, opt_dc):
        """the same quantities in float64 without any of the f16 roundings: what the bound is a bound against -> (score[p], c[p], e[p])"""
        d = self.d
        K, L = d.K, d.L[j]
        wl = K + L
        idx = pos - (wl - 1) + np.arange(wl)
        xn = np.where(idx >= 0, stream[np.clip(idx, 0, len(stream) - 1)], 0).astype(np.float64)
        dc = float(np.sum(xn[K - L:K + L]) / (2.0 * L)) if opt_dc else 0.0
        x0 = xn - 0.98 * dc
        if d.iq:
            ws = self.ws[ora_scan.TEMPLATES[j][7]].astype(np.float64)
            xf = np.convolve(x0, ws)[:wl]
            taps = len(ws)
            xf[:taps - 1] -= 0.98 * dc * np.array([np.sum(ws[i + 1:]) for i in range(taps - 1)])
        else:
            xf = x0
        P = np.concatenate([[0.0], np.cumsum(xf * xf)])
        c = np.correlate(xf, self.match[j].astype(np.float64), mode="valid")[:K + 1]
        e = P[L:L + K + 1] - P[:K + 1]
        return np.where(e > 0, np.abs(c) / np.sqrt(np.maximum(e, 1e-300)), 0.0), c, e

    def run(self, j, stream, pos, opt_dc, detail=False):
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
        # (the kernel sums 16 squares per thread, scans the wave, adds the waves: three short levels — a float64 sum rounded once stands for it; C_P covers either)
        P = np.concatenate([[0], np.cumsum((xfh * xfh).astype(np.float64))]).astype(F)
        c = np.correlate(xfh, self.match[j].astype(H).astype(F), mode="valid")[:K + 1].astype(F)     # c[p'] = sum_k match[k] xf[p' + k]
        e = (P[L:L + K + 1] - P[:K + 1]).astype(F)
        X = float(np.max(np.abs(xn[:wl] - F(0.98) * dc))) if wl else 0.0
        kx, beta = self.kappa(j) * X, C_P * float(P[-1])
        ok = e > 0
        es = np.maximum(e, 1e-30)
        plain = np.where(ok, np.abs(c) / np.sqrt(es), 0)
        bound = np.where(ok, (np.abs(c) + kx) / np.sqrt(es) + beta / es, np.inf)          # (no energy under the template: the pair goes to the exact kernel)
        pa = int(np.argmax(np.abs(c)))
        mv = float(c[pa] / math.sqrt(max(float(e[pa]), 1e-30)))
        out = dict(smax=float(bound.max()), smax_plain=float(plain.max()), mv=mv, mp=(-4 if pa in (0, K) else L - 1 + pa), dc=float(dc))
        if detail:
            out.update(score=plain, bound=bound, c=c, e=e, X=X, E_win=float(P[-1]))
        return out
