"""oracle/ora_fsk.py — TEST INFRASTRUCTURE: numpy restatement of the reference's 2-FSK modem (utils/fsk.c).

  fsk_create_core     constants Ts, N, Ndft, Nmem, tc, estimator bins                          fsk.c:114-201
  fsk_demod_freq_est  half-overlapped Hann FFTs, |X| smoothing, peak / mask estimators          fsk.c:438-590
  fsk_demod_core      down-conversion with the float oscillator recurrence, integrators, fine
                      timing, nin control, interpolated soft decisions, Eb/N0                    fsk.c:593-836
float32 arithmetic is kept where the reference uses it (separately rounded products, serial sums in the reference's
order); the FFT is numpy's (only the peak positions depend on it).  Pinned against recordings of the compiled reference
in tests/golden/fsk_*.npz (tests/test_oracle_fsk.py).  Only tests/ may import this module.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import math

import numpy as np

F = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _f in ("cosf", "sinf", "atan2f", "log10f"):
    getattr(_libm, _f).restype = ctypes.c_float
    getattr(_libm, _f).argtypes = [ctypes.c_float] * (2 if _f == "atan2f" else 1)


def cosf(x):
    """The C library's cosf — the very function the reference calls (comp_exp_j, comp_prim.h:95); a value one ulp off would
    grow to ~2e-4 over a modem frame through the oscillator recurrence."""
    return F(_libm.cosf(float(x)))


def sinf(x):
    return F(_libm.sinf(float(x)))


def _cmult(ar, ai, br, bi):
    return (ar * br - ai * bi).astype(F), (ar * bi + ai * br).astype(F)


class FskModem:
    def __init__(self, Fs, Rs, *, P=8, nsym=50, lower=None, upper=None, mask=0, fmt=2):
        self.Fs, self.Rs, self.P, self.nsym, self.fmt = Fs, Rs, P, nsym, fmt
        Ndft = F(Fs) / F(0.1 * Rs)
        self.Ndft = Ndft = int(2.0 ** math.ceil(math.log2(float(Ndft))))
        self.Ts = Ts = Fs // Rs
        self.N = Ts * nsym
        self.Nmem = self.N + 2 * Ts
        self.tc = F(0.95 * float(F(Ndft)) / Fs)
        self.nin = self.N
        self.est_type = 1 if mask else 0
        self.fs_tx = mask if mask else 100
        lower = (-Fs // 2 if fmt != 1 else 0) if lower is None else lower
        upper = Fs // 2 if upper is None else upper
        cdiv = lambda a, b: int(a / b)                                   # C integer division (truncation)
        self.st = max(cdiv(lower * Ndft, Fs) + Ndft // 2, 0)
        self.en = min(cdiv(upper * Ndft, Fs) + Ndft // 2, Ndft)
        self.f_zero = cdiv(int(0.75 * Rs) * Ndft, Fs)
        self.hann = np.array([F(0.5 - 0.5 * float(cosf(F(2.0 * math.pi * float(F(i)) / float(F(Ndft - 1)))))) for i in range(Ndft)], F)
        self.Sf = np.zeros(Ndft, F)
        self.phi = [np.array([1.0, 0.0], F), np.array([1.0, 0.0], F)]
        self.f_dc = np.zeros((2, self.Nmem, 2), F)
        self.norm_rx_timing = F(0)
        self.ppm = F(0)
        self.snr_est = F(0)
        self.EbNodB = F(0)
        # timing oscillator: phi_ft used, then advanced (fsk.c:682-703)
        a_ft = F(2 * math.pi * float(F(Rs) / F(P * Rs)))
        d = (cosf(a_ft), sinf(a_ft))
        W = (nsym + 1) * P
        self.phi_ft = np.zeros((W, 2), F)
        pr, pi = F(1), F(0)
        for i in range(W):
            self.phi_ft[i] = (pr, pi)
            pr, pi = F(pr * d[0] - pi * d[1]), F(pr * d[1] + pi * d[0])
        b = int(round(float(F(1) * F(self.fs_tx) * F(Ndft) / F(Fs)))) - 1
        self.mask_idx = sorted(set([0, 1, 2] + ([b, b + 1, b + 2] if 0 <= b and b + 2 < Ndft else [])))
        self.len_mask = b + 3

    def convert(self, raw):
        if self.fmt == 2:
            return (raw[0::2].astype(F) / F(1000)), (raw[1::2].astype(F) / F(1000))
        if self.fmt == 1:
            return raw.astype(F) / F(1000), np.zeros(len(raw), F)
        return ((raw[0::2].astype(F) - F(127.0)) / F(128.0)), ((raw[1::2].astype(F) - F(127.0)) / F(128.0))

    def freq_est(self, xr, xi):
        Ndft, nin = self.Ndft, len(xr)
        numffts = nin // (Ndft // 2) - 1
        omt = F(1) - self.tc
        for j in range(numffts):
            a = j * Ndft // 2
            X = np.fft.fftshift(np.fft.fft((self.hann * xr[a:a + Ndft]).astype(np.float64) + 1j * (self.hann * xi[a:a + Ndft]).astype(np.float64)))
            mag = np.sqrt((X.real.astype(F) * X.real.astype(F) + X.imag.astype(F) * X.imag.astype(F)).astype(F)).astype(F)
            self.Sf = ((self.Sf * omt).astype(F) + (mag * self.tc).astype(F)).astype(F)
        work = self.Sf.copy()
        freqi = []
        for _ in range(2):
            seg = work[self.st:self.en]
            imax = self.st + int(np.argmax(seg)) if len(seg) and seg.max() > 0 else 0
            work[max(imax - self.f_zero, 0):min(imax + self.f_zero, Ndft)] = 0
            freqi.append(imax - Ndft // 2)
        freqi.sort()
        f_est = [F(F(k) * (F(self.Fs) / F(Ndft))) for k in freqi]
        if self.est_type:
            best, b_max = F(0), self.st
            for b in range(self.st, self.en - self.len_mask):
                c = F(0)
                for i in self.mask_idx:
                    c = F(c + self.Sf[b + i])
                if c > best:
                    best, b_max = c, b
            foff = F(int((b_max - Ndft // 2) * self.Fs / Ndft))
            f_est = [F(foff + F(m * self.fs_tx)) for m in range(2)]
        return f_est

    def frame(self, raw):
        """One fsk_demod_sd() call on exactly self.nin samples -> (soft decisions, record)."""
        Ts, P, nsym, N, Nmem = self.Ts, self.P, self.nsym, self.N, self.Nmem
        xr, xi = self.convert(raw)
        nin = len(xr)
        f_est = self.freq_est(xr, xi)
        nold = Nmem - nin
        self.f_dc[:, :nold] = self.f_dc[:, Nmem - nold:].copy()
        for m in range(2):
            ang = F(2 * math.pi * float(F(f_est[m]) / F(self.Fs)))
            dr, di = cosf(ang), sinf(ang)
            pr, pi = self.phi[m]
            ph = np.zeros((nin, 2), F)
            for j in range(nin):
                pr, pi = F(pr * dr - pi * di), F(pr * di + pi * dr)
                ph[j] = (pr, pi)
            fr, fi = _cmult(xr, xi, ph[:, 0], (-ph[:, 1]).astype(F))
            self.f_dc[m, nold:, 0], self.f_dc[m, nold:, 1] = fr, fi
            av = F(np.sqrt(F(pr * pr + pi * pi)))
            self.phi[m] = np.array([pr / av, pi / av], F)
        W = (nsym + 1) * P
        st = (np.arange(W) * Ts) // P
        f_int = np.zeros((2, W, 2), F)
        for j in range(Ts):
            f_int = (f_int + self.f_dc[:, st + j, :]).astype(F)
        ft1 = np.zeros(W, F)
        for m in range(2):
            ft1 = (ft1 + (f_int[m, :, 0] * f_int[m, :, 0] + f_int[m, :, 1] * f_int[m, :, 1]).astype(F)).astype(F)
        tcr = np.cumsum((ft1 * self.phi_ft[:, 0]).astype(F), dtype=F)[-1]
        tci = np.cumsum((ft1 * self.phi_ft[:, 1]).astype(F), dtype=F)[-1]
        norm = F(float(F(_libm.atan2f(float(tci), float(tcr)))) / (2 * math.pi))
        rx_timing = F(norm * F(P))
        d_norm = F(norm - self.norm_rx_timing)
        self.norm_rx_timing = norm
        if abs(float(d_norm)) < .2:
            appm = F(1e6 * float(d_norm) / float(F(nsym)))
            self.ppm = F(.9 * float(self.ppm) + .1 * float(appm))
        nin_next = N + Ts // 2 if norm > 0.25 else (N - Ts // 2 if norm < -0.25 else N)
        low, high = int(math.floor(float(rx_timing))), int(math.ceil(float(rx_timing)))
        fract = F(rx_timing - F(low)); omf = F(F(1) - fract)
        sp = (np.arange(nsym) + 1) * P
        tmax = []
        for m in range(2):
            tr = ((omf * f_int[m, sp + low, 0]).astype(F) + (fract * f_int[m, sp + high, 0]).astype(F)).astype(F)
            ti = ((omf * f_int[m, sp + low, 1]).astype(F) + (fract * f_int[m, sp + high, 1]).astype(F)).astype(F)
            tmax.append((tr * tr + ti * ti).astype(F))
        sd = (np.sqrt(tmax[0]).astype(F) - np.sqrt(tmax[1]).astype(F)).astype(F)
        mx = np.maximum(tmax[0], tmax[1])
        stdebno = np.cumsum(mx, dtype=F)[-1]
        meanebno = np.cumsum(np.sqrt(mx).astype(F), dtype=F)[-1]
        meanebno = F(meanebno / F(nsym))
        stdebno = F(F(stdebno / F(nsym)) - F(meanebno * meanebno))
        stdebno = F(math.sqrt(float(stdebno))) if stdebno > 0 else F(0)
        self.EbNodB = F(F(-6) + F(20) * F(_libm.log10f(float(F((1e-6 + float(meanebno)) / (1e-6 + float(stdebno)))))))
        self.snr_est = F(.5 * float(self.snr_est) + .5 * float(self.EbNodB))
        rec = dict(nin=nin, nin_next=nin_next, f_est=(float(f_est[0]), float(f_est[1])), norm_rx_timing=float(norm), ppm=float(self.ppm),
                   EbNodB=float(self.EbNodB), snr_est=float(self.snr_est))
        self.nin = nin_next
        return sd, rec

    def run(self, raw):
        """The `while (fread(fsk_nin))` loop of fsk_demod.c:279 over an in-memory capture."""
        per = 1 if self.fmt == 1 else 2
        pos, n = 0, len(raw) // per
        sds, recs = [], []
        while pos + self.nin <= n:
            sd, rec = self.frame(raw[per * pos:per * (pos + self.nin)])
            pos += rec["nin"]
            sds.append(sd); recs.append(rec)
        return np.array(sds), recs
