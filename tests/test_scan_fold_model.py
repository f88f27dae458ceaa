"""The algebra behind the scanner's one-pass front end, checked in numpy — no GPU, no library.

dft_detect mixes and decimates (x - mean) where mean is the IQ average of the PREVIOUS 1/32 s window (scan/dft_detect.c:539-588, 1085-1101).  The kernels
compute the raw sum and take the means off per OUTPUT:

    y[m] = sum_k w[k] (x[n] - mean(n)) ex[n]  =  sum_k w[k] x[n] ex[n]  -  sum_q mean(block m-(Q-1)+q) * Eblk(m, q),      n = D (m+1) - T + k
    Eblk(m, q) = sum_r W_q[r] ex[D (m-(Q-1)+q) + r],   W_q[r] = front-padded tap D q + r,   E[m] = sum_q Eblk(m, q)

with one mean for all Q blocks of an output except for the Q-1 outputs behind a change of the mean, which get
    corr[i] = (mean_new - mean_old) * sum_{q < Q-1-i} Eblk(m, q)        (k_scan_dc_edges)
on top of  - mean_new * E[m]  (k_scan_if).  This file asserts exactly that identity on random data, window edges at arbitrary blocks.
"""
import numpy as np


def test_fold_identity_with_edge_corrections():
    rng = np.random.default_rng(5)
    D, Q = 50, 7
    H = Q - 1
    T = 343                                              # taps of the 2.4 Msps -> 48 kHz decimator
    w = rng.standard_normal(T) * 0.05
    wpad = np.concatenate([np.zeros(Q * D - T), w])      # front padded to Q blocks: W_q[r] = wpad[D q + r]
    nblocks, Bw = 40, 9                                  # IQ-DC windows of 9 blocks (the real ones are 1500): edges everywhere
    n = nblocks * D
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n) + (0.7 - 0.4j)
    f0 = 0.0123
    ex = np.exp(2j * np.pi * f0 * np.arange(n))
    off = 4                                              # the launch starts 4 blocks into a window
    nwin = (off + nblocks + Bw - 1) // Bw
    means = rng.standard_normal(nwin + 1) * 0.3 + 1j * rng.standard_normal(nwin + 1) * 0.3      # means[k + 1]: window k of the launch; means[0]: the window before
    win_of_block = lambda j: (j + off) // Bw             # j >= 0
    mean_of_sample = np.array([means[win_of_block(i // D) + 1] for i in range(n)])

    def direct(m):                                       # what the reference computes (samples before the launch: not part of this check)
        idx = np.arange(D * (m - H), D * (m + 1))
        return np.sum(wpad * (x[idx] - mean_of_sample[idx]) * ex[idx])

    def eblk(m, q):
        idx = np.arange(D * (m - H + q), D * (m - H + q + 1))
        return np.sum(wpad[D * q:D * (q + 1)] * ex[idx])

    for m in range(H, nblocks):
        idx = np.arange(D * (m - H), D * (m + 1))
        raw = np.sum(wpad * x[idx] * ex[idx])                                   # k_mix_decimate50r
        kw = win_of_block(m)
        rem = (m + off) - kw * Bw                                               # blocks since the window's start
        E = sum(eblk(m, q) for q in range(Q))
        y = raw - means[kw + 1] * E                                             # k_scan_if: the window's mean times E
        if rem < H:                                                             # + k_scan_dc_edges: the first H - rem blocks ran under the mean before
            y += (means[kw + 1] - means[kw]) * sum(eblk(m, q) for q in range(H - rem))
        assert abs(y - direct(m)) < 1e-10 * (1 + abs(y)), (m, kw, rem)
