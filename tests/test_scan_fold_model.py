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


def test_fold_bookkeeping_across_calls():
    """The same identity with the state the kernels carry from call to call, restated step by step (sonde_scan.cpp's one-pass branch): k_dc_seg_means' table of
    window means (`seg`, the running window's sum carried over, the mean of the window before the call's first one kept at its value of the call's START),
    k_scan_dc_edges' corrections per (window of the call, output behind its start) and k_scan_if's fold — calls of ragged lengths, some shorter than the
    decimator's history, some starting on a change of the mean, some a few blocks behind one.  Against the reference's definition: every sample has the mean
    of the 1/32 s window BEFORE its own taken off (dft_detect.c:539-588), zero in the first window."""
    rng = np.random.default_rng(9)
    D, Q = 50, 7
    H = Q - 1
    T = 343
    w = rng.standard_normal(T) * 0.05
    wpad = np.concatenate([np.zeros(Q * D - T), w])
    Bw = 9                                               # blocks per IQ-DC window
    calls = [3, 1, 2, 3, 11, 9, 18, 5, 1, 7, 27, 4, 9, 9, 2]
    nblocks = sum(calls)
    n = nblocks * D
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n) + (0.9 + 0.5j) + 0.3 * np.sin(np.arange(n) / 700.0)
    ex = np.exp(2j * np.pi * 0.0031 * np.arange(n))
    # the reference: window v runs under the mean of window v - 1 (0 for v = 0)
    nwin = (nblocks + Bw - 1) // Bw
    wmean = [0j] + [x[v * Bw * D:(v + 1) * Bw * D].mean() for v in range(nwin)]          # wmean[v] = mean in effect during window v
    xs = np.concatenate([np.zeros(H * D, complex), x - np.repeat(wmean[:nwin], Bw * D)[:n]])   # samples in front of the stream: zeros
    xr = np.concatenate([np.zeros(H * D, complex), x])
    exs = np.concatenate([np.exp(2j * np.pi * 0.0031 * np.arange(-H * D, 0)), ex])

    def blocks(arr, m):                                  # the Q blocks of output m (block m last)
        return arr[D * m:D * (m + H + 1)]                # (shifted by the H blocks of zeros in front)

    def eblk(m, q):
        return np.sum(wpad[D * q:D * (q + 1)] * exs[D * (m + q):D * (m + q + 1)])

    direct = np.array([np.sum(wpad * blocks(xs, m) * blocks(exs, m)) for m in range(nblocks)])
    raw = np.array([np.sum(wpad * blocks(xr, m) * blocks(exs, m)) for m in range(nblocks)])      # k_mix_decimate50r (its P tail carries the blocks of earlier calls)
    E = np.array([sum(eblk(m, q) for q in range(Q)) for m in range(nblocks)])

    dc_avg, dc_prev, cx, g0 = 0j, 0j, 0j, 0              # scanner state: mean in effect, mean of the window before, sum of the window in progress, blocks so far
    out = np.zeros(nblocks, complex)
    for nb in calls:
        off = g0 % Bw
        nseg, ncomplete = (off + nb + Bw - 1) // Bw, (off + nb) // Bw
        # k_dc_rows_to_segments + k_dc_seg_means
        seg, mean, prev = [], dc_avg, dc_prev
        for k in range(nseg + 1):
            seg.append(mean)
            if k >= nseg:
                continue
            lo, hi = max(0, k * Bw - off), min(nb, (k + 1) * Bw - off)
            cx += x[(g0 + lo) * D:(g0 + hi) * D].sum()
            if k < ncomplete:
                prev, mean, cx = mean, cx / (Bw * D), 0j
        dc_prev_start, dc_avg, dc_prev = dc_prev, mean, prev           # (the edge kernel reads the START value: two arrays in the scanner)
        # k_scan_dc_edges
        corr = np.zeros((nseg, H), complex)
        for k in range(nseg):
            for i in range(H):
                m = k * Bw - off + i
                if 0 <= m < nb:
                    corr[k, i] = (seg[k] - (seg[k - 1] if k > 0 else dc_prev_start)) * sum(eblk(g0 + m, q) for q in range(H - i))
        # k_scan_if's fold
        for m in range(nb):
            kw, rem = divmod(m + off, Bw)
            y = raw[g0 + m] - seg[kw] * E[g0 + m]
            if rem < H:
                y += corr[kw, rem]
            out[g0 + m] = y
        g0 += nb
    assert np.abs(out - direct).max() < 1e-9 * (1 + np.abs(direct).max())
