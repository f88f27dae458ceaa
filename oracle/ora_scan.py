"""oracle/ora_scan.py — TEST INFRASTRUCTURE: numpy restatement of the reference scanner's correlation stage.

Restates, from FM-stream samples on, what scan/dft_detect.c computes per correlation window and template:
  templates      rs_hdr[] rows, Gaussian-pulse header template incl. the "longest header" quirk     dft_detect.c:172-191,1227-1258
  transform      dft_raw(): bit reversal + radix-2 DIT, stage twiddle by float recurrence           dft_detect.c:285-322
  getCorrDFT     window, dc over the last 2L samples, X[0] -= N*dc*0.98, FM low-pass and matched
                 filter as spectral products, peak of Re(cx)^2, edge reject, 2-norm over L samples  dft_detect.c:357-443
  headcmp        float bit clock, hard bits of the header against the template                      dft_detect.c:821-905
It is pinned against the compiled reference through tests/golden/scan_*.npz (tests/test_oracle_scan.py) and checks
the GPU kernels where the harness cannot travel.  float32 arithmetic is kept wherever the reference uses it.
Only tests/ may import this module.
"""
from __future__ import annotations

import math

import numpy as np

N, LOG2N = 8192, 13
F = np.float32

# (baud, header, BT, thres, herrs, type, tn, lpFM, lpIQ) — rs_hdr[0..15]
TEMPLATES = [
    (2500, "10011010100110010101101001010101", 1.0, 0.65, 2, "DFM9", 2, 0, 1),
    (4800, "0000100001101101010100111000100001000100011010010100100000011111", 0.5, 0.70, 2, "RS41", 3, 0, 1),
    (4800, "10100110011001101001" * 2 + "1010011001100110100110101010100110101001", 0.5, 0.70, 3, "RS92", 4, 0, 1),
    (4800, "0101011000001000" "0001110010010111" "0001101010100111" "0011110100111110", 1.0, 0.60, 8, "LMS6", 8, 0, 1),
    (4800, "0000000001" "0101010101" "0001001001" "0001001001", 0.5, 0.80, 2, "IMET5", 24, 0, 1),
    (9616, "0010100111" "0010100111" "0001001001" "0010010101", 1.0, 0.70, 2, "MK2LMS", 18, 1, 2),
    (9608, "1001100110010100110010011001" "1010", 1.0, 0.76, 2, "M10", 5, 1, 2),
    (2400, "110011001101001101001101010100101010110010101010", 1.0, 0.70, 2, "MEISEI", 9, 0, 2),
    (4800, "10100110010110101001" "10010101011010010101" "10101001010101010101" "10011001010110101001", 1.0, 0.70, 2, "RD94RD41", 10, 0, 1),
    (2400, "1001100110011001" "1001101010101010", 1.5, 0.80, 2, "MRZ", 12, 0, 1),
    (1200, "10101010" "10101010" "10110100" "00101011", 1.0, 0.65, 2, "MTS01", 13, 0, 0),
    (5800, "01010101010101010101010101010101", 1.5, 0.80, 2, "C34C50", 15, 0, 2),
    (4800, "10101010" * 3 + "00101101" "11010100", 1.0, 0.65, 2, "WXR301", 16, 0, 3),
    (5000, "10101010" * 3 + "11000001" "10010100", 1.0, 0.65, 2, "WXRPN9", 17, 0, 3),
    (9600, "0000" "11110000111100001111000011110000" "1111" "0000" "10101100110010101100101010101100" "1111", 1.0, 0.80, 2, "IMET1AB", 29, 1, 3),
    (9600, "11110000111100001111000011110000" * 2, 0.5, 0.80, 4, "IMETafsk", 25, 1, 1),
]
DISABLED = (11, 14)          # -DNOC34C50 -DNOIMET1AB (scan/Makefile:1)


def _cmul(a, b):
    """complex64 product with separately rounded float32 multiplies/adds (no fused operations)."""
    ar, ai, br, bi = a.real.astype(F), a.imag.astype(F), b.real.astype(F), b.imag.astype(F)
    return ((ar * br - ai * bi) + 1j * (ar * bi + ai * br)).astype(np.complex64)


def twiddles():
    """Stage twiddles of dft_raw: w1 = 1; per butterfly column w1 *= cexp(-i pi / 2^s), all in float."""
    tws = []
    for s in range(LOG2N):
        l2 = 1 << s
        e = np.complex64(np.exp(-1j * math.pi / float(F(l2))))
        w = np.complex64(1.0)
        out = np.zeros(l2, np.complex64)
        for j in range(l2):
            out[j] = w
            w = _cmul(np.array([w]), np.array([e]))[0]
        tws.append(out)
    return tws


_TW = None
_BR = np.array([int("{:013b}".format(i)[::-1], 2) for i in range(N)])


def dft_ref(x):
    """The reference's transform (not the exact DFT: it carries the twiddle recurrence's drift)."""
    global _TW
    if _TW is None:
        _TW = twiddles()
    z = np.asarray(x).astype(np.complex64)[_BR].copy()
    for s in range(LOG2N):
        l2 = 1 << s
        z = z.reshape(N >> (s + 1), 2 * l2)
        t = _cmul(z[:, l2:], _TW[s][None, :])
        a = z[:, :l2].copy()
        z = np.concatenate([(a + t).astype(np.complex64), (a - t).astype(np.complex64)], axis=1).reshape(N)
    return z


def lowpass_taps(f, taps):
    """lowpass_init (dft_detect.c:662-694): Blackman x sinc, taps stored as float, 1-norm accumulated in double."""
    f = F(f)
    if taps % 2 == 0:
        taps += 1
    twof = F(2) * f
    ws = np.zeros(taps, F)
    norm = 0.0
    for n in range(taps):
        w = 7938 / 18608.0 - 9240 / 18608.0 * math.cos(2 * math.pi * n / (taps - 1)) + 1430 / 18608.0 * math.cos(4 * math.pi * n / (taps - 1))
        x = float(twof * F(n - (taps - 1) // 2))
        sinc = 1.0 if x == 0 else math.sin(math.pi * x) / (math.pi * x)
        ws[n] = F(w * (float(twof) * sinc))
        norm += float(ws[n])
    return np.array([F(float(v) / norm) for v in ws], F)


def _pulse(t, sigma):
    q = lambda x: 0.5 - 0.5 * math.erf(x / 1.4142135624)
    return q((t - 0.5) / sigma) - q((t + 0.5) / sigma)


def match_template(bits, hlen_max, spb, bt, L):
    spb = F(spb)
    sigma = math.sqrt(math.log(2)) / (2 * math.pi * float(F(bt)))
    m = np.zeros(L, F)
    chars = bits + "\0"
    for i in range(L):
        pos = int(F(i) / spb)
        t = float((F(i) - F(pos) * spb) / spb) - 0.5
        b = ((ord(chars[pos]) & 1) - 0.5) * 2.0 * _pulse(t, sigma)
        if pos > 0:
            b += ((ord(chars[pos - 1]) & 1) - 0.5) * 2.0 * _pulse(t + 1, sigma)
        if pos < hlen_max - 1:
            b += ((ord(chars[pos + 1]) & 1) - 0.5) * 2.0 * _pulse(t - 1, sigma)      # the terminator counts as a 0 bit
        m[i] = F(b)
    nm = F(math.sqrt(float(np.sum(m.astype(np.float64) ** 2))))
    return (m / nm).astype(F)


class ScanDesign:
    """init_buffers() of dft_detect.c:995-1285 for an IF rate `sr` (after decimation)."""

    def __init__(self, sr, iq=True):
        self.sr, self.iq = sr, iq
        self.spb = [F(sr) / F(t[0]) for t in TEMPLATES]
        self.hlen = [len(t[1]) for t in TEMPLATES]
        self.L = [int(float(F(self.hlen[j]) * self.spb[j]) + 0.5) for j in range(16)]
        act = [j for j in range(16) if j not in DISABLED]
        self.active = act
        hmax = max(self.hlen[j] for j in act)
        L2 = 2 * max(self.L[j] for j in act)
        p2 = 1
        while p2 < 3 * L2:
            p2 <<= 1
        while p2 < 0x2000:
            p2 <<= 1
        assert p2 == N
        self.K, self.delay = N - L2, L2 // 16
        self.lpfm_taps = 0
        self.WS = [None, None]
        if iq:
            taps = int(4 * sr / 2e3)
            taps += (taps % 2 == 0)
            for k, bw in enumerate((4e3, 10e3)):
                w = lowpass_taps(F(bw) / F(sr), taps)
                self.lpfm_taps = len(w)
                m = np.zeros(N, F); m[:len(w)] = w
                self.WS[k] = dft_ref(m)
        self.Fm = {}
        for j in act:
            match = match_template(TEMPLATES[j][1], hmax, self.spb[j], TEMPLATES[j][2], self.L[j])
            m = np.zeros(N, F); m[:self.L[j]] = match[::-1]
            self.Fm[j] = dft_ref(m)

    def corr(self, j, stream, pos, opt_dc):
        """getCorrDFT for template j on a window ending at sample_out = pos; stream[i] = buf_fm sample i (zeros before 0)."""
        K, L = self.K, self.L[j]
        idx = pos - (K + L - 1) + np.arange(K + L)
        xn = np.zeros(N, F)
        xn[:K + L] = np.where(idx >= 0, stream[np.clip(idx, 0, len(stream) - 1)], 0).astype(F)
        X = dft_ref(xn)
        dc = F(0)
        if opt_dc:
            s = F(0)
            for v in xn[K - L:K + L]:
                s = F(s + v)
            dc = F(float(s) / (2.0 * float(F(L))))
            X[0] = np.complex64(X[0] - F(float(F(N) * dc) * 0.98))
        if self.iq:
            X = _cmul(X, self.WS[TEMPLATES[j][7]])
        if opt_dc or self.iq:
            xn = (dft_ref(np.conj(X)).real / F(N)).astype(F)
        cx = dft_ref(np.conj(_cmul(X, self.Fm[j]))).real.astype(F)
        seg = cx[L - 1:K + L]
        if not np.any(seg * seg > 0):
            # digital silence: the reference's loop (`re_cx*re_cx > mx2`, dft_detect.c:417) leaves mp = -1, which is no edge value — it goes on, stores 0 / (a norm
            # read in front of the array) and the WRAPPED position, and `mv_pos > mv0_pos` (:1521) then fails for the next window's hit
            return dict(mp=-1, mv=0.0, mpos=(pos - (K + L - 1) - 1 - (self.lpfm_taps // 2 if self.iq else 0)) & 0xFFFFFFFF, dc=float(dc))
        mp = L - 1 + int(np.argmax(seg * seg))
        if mp == L - 1 or mp == K + L - 1:
            return dict(mp=-4, dc=float(dc))
        xnorm = math.sqrt(float(np.sum((xn[mp - L + 1:mp + 1] * xn[mp - L + 1:mp + 1]).astype(np.float64))))
        mv = F(float(cx[mp]) / (xnorm * N))
        mpos = pos - (K + L - 1) + mp - (self.lpfm_taps // 2 if self.iq else 0)
        return dict(mp=mp, mv=float(mv), mpos=mpos, dc=float(dc))

    def headcmp(self, j, stream, mpos, inv, dc):
        """Header bit errors (read_bufbit / headcmp, dft_detect.c:821-905)."""
        spb, hl = self.spb[j], self.hlen[j]
        mvp = mpos + 1 - int(float(F(hl) * spb))
        rcount, grenze, errs = 0, F(0), 0
        for b in range(hl):
            grenze = F(grenze + spb)
            s = 0.0
            while True:
                p = rcount + mvp
                s += float(F((stream[p] if 0 <= p < len(stream) else 0.0)) - F(dc))
                rcount += 1
                if not (F(rcount) < grenze):
                    break
            bit = 1 if s >= 0 else 0
            errs += ((bit ^ int(inv)) != (ord(TEMPLATES[j][1][b]) & 1))
        return errs
