"""CPU: the prefilter's obligation (tests/scan_pre_model.py = k_scan_pre's arithmetic in numpy) against the pinned restatement of the
reference's getCorrDFT (oracle/ora_scan.py): on FM streams with headers of several types, both polarities, a dc offset, noise and a header
cut by the window edge, every (window, template) whose reference score reaches its threshold (less 5e-4) is a prefilter candidate
(bound > thres - MARGIN, tests/scan_pre_model.py), the prefilter's own score is within 1e-3 of the reference's wherever both pick the same peak, the prefilter
is selective (a few per cent of the pairs are candidates), and — round 6 — the DERIVED bound of DESIGN.md §4.6b holds position by position against the same
quantities in float64, also where the window's maximum is 30 times the rms under the template."""
import numpy as np
import pytest

from scan_pre_model import PrefilterModel, MARGIN, U16


@pytest.fixture(scope="module")
def setup():
    from oracle import ora_scan
    d = ora_scan.ScanDesign(48000, iq=True)
    return ora_scan, d, PrefilterModel(d)


def _stream(pre, rng, n, dc, noise, plants):
    s = (noise * rng.standard_normal(n)).astype(np.float32) + np.float32(dc)
    for j, at, sgn, amp in plants:
        w = pre.match[j] / np.abs(pre.match[j]).max() * amp
        seg = s[at:at + len(w)]
        seg += (sgn * w[:len(seg)]).astype(np.float32)
    return s


@pytest.mark.parametrize("opt_dc,dc,noise", [(1, 0.11, 0.02), (0, 0.0, 0.02), (1, -0.3, 0.05), (0, 0.02, 0.08)])
def test_prefilter_is_a_superset_and_close(setup, opt_dc, dc, noise):
    ora_scan, d, pre = setup
    rng = np.random.default_rng(int(1000 * abs(dc)) + opt_dc)
    n = 46000
    K = d.K
    first = K - 4 - d.delay - 1
    plants = [(1, 9000, 1, 0.08), (0, 25000, -1, 0.08), (6, 40000, 1, 0.06), (9, 30500, 1, 0.03),
              (1, first + (K - 4) - 300, -1, 0.08)]                                   # the last one straddles a window's end
    stream = _stream(pre, rng, n, dc, noise, plants)
    worst, ncand, nall, hits = 0.0, 0, 0, 0
    for pos in range(first, n, K - 4):
        for j in d.active:
            ex = d.corr(j, stream, pos, opt_dc)
            pr = pre.run(j, stream, pos, opt_dc)
            thres = ora_scan.TEMPLATES[j][3]
            cand = pr["smax"] > thres - MARGIN
            if pos - (K + d.L[j] - 1) >= 0:                        # (selectivity is counted where the window lies inside the stream: the zeros in front of a stream's first
                nall += 1                                          # sample have no energy, and a span without energy is — rightly — never ruled out by rounding arguments:
                ncand += cand                                      # the first window of a channel goes to the exact kernel, once)
            if ex["mp"] > 0:
                assert pr["smax"] > abs(ex["mv"]) - 1e-3, (pos, j, ex, pr)               # upper bound up to the f16 rounding
                if abs(ex["mv"]) > thres - 5e-4:
                    assert cand, (pos, j, ex, pr)
                    hits += abs(ex["mv"]) > thres
                if ex["mp"] == pr["mp"]:
                    worst = max(worst, abs(ex["mv"] - pr["mv"]))
            assert abs(ex["dc"] - pr["dc"]) < 1e-5
    assert hits >= 3 and worst < 1e-3 and ncand < 0.1 * nall, (hits, worst, ncand, nall)


@pytest.mark.parametrize("burst", [0.0, 0.8])
@pytest.mark.parametrize("opt_dc,dc,noise", [(1, 0.11, 0.02), (0, 0.0, 0.005), (1, -0.3, 0.05)])
def test_prefilter_error_stays_inside_the_derived_bound(setup, opt_dc, dc, noise, burst):
    """DESIGN.md §4.6b, position by position: |score in f16 / f32 arithmetic - score in float64| <= 3 u + L 2^-24 + kappa X / sqrt(e) + beta / e — the terms the
    candidate test adds to the f16 score (the two that scale with the signal) or takes off the threshold (the constant).  burst: a stretch of full-scale FM noise
    (the discriminator's output is uniform in +-0.8 where there is no carrier) beside weak headers — the window's maximum is then 30 x the rms under a template, the
    case a flat margin would not cover."""
    ora_scan, d, pre = setup
    rng = np.random.default_rng(77 + int(100 * noise) + opt_dc)
    n = 30000
    K = d.K
    first = K - 4 - d.delay - 1
    stream = _stream(pre, rng, n, dc, noise, [(1, 9000, 1, 0.08), (0, 14000, -1, 0.03), (6, 20000, 1, 0.06)])
    if burst:
        stream[11000:12500] = (np.float32(dc) + rng.uniform(-burst, burst, 1500)).astype(np.float32)
    worst_ratio, worst_plain, rho_max = 0.0, 0.0, 0.0
    for pos in range(first, n, K - 4):
        for j in d.active:
            pr = pre.run(j, stream, pos, opt_dc, detail=True)
            ideal, _, e64 = pre.ideal(j, stream, pos, opt_dc)
            L = d.L[j]
            ok = (pr["e"] > 0) & (e64 > 1e-12)
            err = np.abs(pr["score"] - ideal)[ok]
            allowed = (3 * U16 + L * 2.0 ** -24 + (pr["bound"] - pr["score"]))[ok]
            assert np.all(err <= allowed), (pos, j, float((err - allowed).max()))
            worst_ratio = max(worst_ratio, float((err / allowed).max()))
            worst_plain = max(worst_plain, float(err.max()))
            rho_max = max(rho_max, float((pr["X"] * np.sqrt(L) / np.sqrt(e64[ok])).max())) if np.any(ok) else rho_max
            # and what it is for: a position whose exact score reaches the threshold is inside the candidate test
            thres = ora_scan.TEMPLATES[j][3]
            assert not np.any((ideal[ok] >= thres) & (pr["bound"][ok] <= thres - MARGIN)), (pos, j)
    assert worst_ratio < 1.0 and (rho_max > 25 if burst else True), (worst_ratio, worst_plain, rho_max)


@pytest.mark.parametrize("n_taps,n_out", [(97, 700), (640, 1200), (1, 40), (16, 64), (17, 33), (1280, 300)])
def test_toeplitz_fragment_tables(n_taps, n_out):
    """The f16 A-fragment tables the matrix-core kernel multiplies with (sonde_scan.cpp toeplitz_frags), contracted on the host exactly the way
    v_mfma_f32_16x16x32_f16 pairs them with the window (lane = row + 16 g, k = 8 g + r), give out[i] = sum_u h[u] x[i + u]: every tap once."""
    import ctypes as C
    from radiosonde_auto_rx_amd.engine import lib
    L = lib()
    L.sonde_scan_toeplitz_model.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    rng = np.random.default_rng(n_taps)
    h = rng.standard_normal(n_taps).astype(np.float32) * 0.1
    x = rng.standard_normal(n_out + n_taps + 40).astype(np.float32)
    out = np.zeros(n_out, np.float32)
    nc = L.sonde_scan_toeplitz_model(h.ctypes.data, n_taps, x.ctypes.data, len(x), out.ctypes.data, n_out)
    assert nc == (((n_taps + 15 + 31) // 32 + 1) & ~1)        # full fragments (32 taps a step), whole blocks of 2 steps, zero padded
    h16, x16 = h.astype(np.float16).astype(np.float64), x.astype(np.float16).astype(np.float64)
    want = np.array([np.dot(h16, x16[i:i + n_taps]) for i in range(n_out)])
    assert np.abs(out - want).max() < 2e-5 * max(1.0, np.abs(want).max())
