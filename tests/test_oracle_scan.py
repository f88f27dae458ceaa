"""CPU: the numpy restatement of the scanner's correlation stage (oracle/ora_scan.py) against values recorded from the
compiled reference (tests/golden/scan_*.npz: oracle/ref_scan_harness.c drives the reference's own getCorrDFT/headcmp).

Pins: template design (incl. the longest-header quirk), the drifting radix-2 transform, dc / FM-low-pass / matched
filter in the spectrum, peak, norm, header bit clock.  Tolerances: indices exact; score 5e-6 (the reference itself is
an -Ofast build); dc 1e-7."""
import numpy as np
import pytest
from golden_cases import SCAN_NAMES, load_scan

LPIQ_STREAM = [1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 0, 2, 3, 3, 3, 1]       # rs_hdr[j].lpIQ (dft_detect.c:172-191)
CASES = [n for n in SCAN_NAMES if "tap_fm" in load_scan(n)]


@pytest.fixture(scope="module")
def design():
    from oracle import ora_scan
    return ora_scan.ScanDesign(48000, iq=True)


def test_design_matches_reference_constants(design):
    g = load_scan(CASES[0])
    assert design.K == g["consts"]["K"] and design.delay == g["consts"]["delay"]
    assert design.L == g["consts"]["L"] and design.lpfm_taps == g["consts"]["lpfm_taps"]


def test_transform_is_the_drifting_one():
    """dft_raw's float twiddle recurrence is off the exact DFT by ~4e-5 — enough to move the 4th decimal of a score."""
    from oracle import ora_scan
    x = np.random.default_rng(1).standard_normal(ora_scan.N).astype(np.float32)
    a, b = ora_scan.dft_ref(x), np.fft.fft(x.astype(np.float64))
    rel = np.abs(a - b).max() / np.abs(b).max()
    assert 3e-6 < rel < 2e-4


@pytest.mark.parametrize("name", CASES)
def test_corr_window_matches_reference(design, name):
    g = load_scan(name)
    from make_golden import SCAN_CASES
    case = SCAN_CASES[name]
    w, jhit, st, first = int(g["tap_w"]), int(g["tap_j"]), int(g["tap_stream"]), int(g["tap_first"])
    pos = int(g["pos"][w])
    stream = np.zeros(first + len(g["tap_fm"]), np.float32)
    stream[first:] = g["tap_fm"]
    single = bool(case["bw"])                                        # --bw: streams 0..2 are the same filter
    checked = 0
    for j in design.active:
        if not (LPIQ_STREAM[j] == st or (single and LPIQ_STREAM[j] < 3 and st < 3)):
            continue
        if pos - (design.K + design.L[j] - 1) < first and first > 0:
            continue                                                 # window starts before the stored segment
        r = design.corr(j, stream, pos, case["dc"])
        assert r["mp"] == g["mp"][w][j], (j, r, g["mp"][w][j])
        assert abs(r["dc"] - g["dc"][w][j]) < 1e-7
        if r["mp"] > 0:
            assert r["mpos"] == g["mpos"][w][j]
            assert abs(r["mv"] - g["mv"][w][j]) < 5e-6, (j, r["mv"], g["mv"][w][j])
            if g["herrs"][w][j] >= 0:
                e = design.headcmp(j, stream, r["mpos"], r["mv"] < 0, r["dc"] if case["dc"] else 0.0)
                assert e == g["herrs"][w][j]
        checked += 1
    assert checked >= 3 and g["herrs"][w][jhit] >= 0


def test_all_zero_first_window_is_the_references_mp_minus_one(design):
    """A stream that begins with digital silence: in the all-zero first window the reference's arg-max leaves mp = -1 and getCorrDFT goes on to store the WRAPPED position
    pos - (K + L - 1) - 1 - lpFMtaps / 2 (dft_detect.c:415-438) — what makes `mv_pos > mv0_pos` (:1521) fail for the header the second window finds.  The restatement's
    window function says the same as the live harness (oracle/_ref/libref_scan.so), template by template."""
    from oracle import bind
    if not bind.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    from tools import synth
    sr = 48_000
    x = np.concatenate([np.zeros(2 * 6000, np.int16), synth.rs41_capture(sr=sr, seconds=1.2, fq=0.0, noise_sigma=0.03, seed=7, t_first=0.005)])
    r = bind.ref_scan_windows(x, sr, iq_mode=1, dc=False, max_win=8)
    assert r["n"] >= 2
    pos0 = int(r["pos"][0])
    zeros = np.zeros(pos0 + 1, np.float32)
    checked = 0
    for j in design.active:
        assert int(r["mp"][0][j]) == -1, (j, r["mp"][0][j])
        o = design.corr(j, zeros, pos0, False)
        assert o["mp"] == -1 and o["mpos"] == int(r["mpos"][0][j]) and o["mpos"] > 0xFFFF0000, (j, o, int(r["mpos"][0][j]))
        checked += 1
    assert checked >= 10
