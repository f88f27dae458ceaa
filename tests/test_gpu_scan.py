"""GPU parity of the batched scanner (include/sonde_scan.h) against the reference's scan/dft_detect.c.

Golden values come from the compiled reference (tools/make_golden.py: the CLI's stdout / exit code, and per-window
score / position / dc / header errors / M10 type bits of every template from oracle/ref_scan_harness.c).
Tolerances:
  window positions, peak indices, header positions, header bit errors, M10 bits, type, exit code ... exact
  text lines (`TYPE: %.4f , %+.1fHz`) ............................................................ identical
  scores mv (the kernels reproduce the reference's radix-2 transform incl. its twiddle recurrence) . 2e-5 abs
  dc (mean of the last 2L window samples) .......................................................... 1e-6 abs
Two modes: `exact=True` runs the reference's transform network for every (window, template) — the per-window parity tests; the default
(what the CLIs and the receivers use) scores every pair with the matrix-core prefilter first and runs the network only where the
prefilter's upper bound comes within 0.03 of the threshold (and for the same template's window before such a pair).  The prefilter tests
below check that this changes nothing that is printed or returned and that every pair it skipped is indeed below its threshold.
"""
import os
import subprocess

import numpy as np
import pytest
from golden_cases import SCAN_NAMES, load_scan, scan_capture

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scanner(case, fq, n_channels=1, **kw):
    from radiosonde_auto_rx_amd.scan import Scanner
    cli = case["cli"]
    tl = float(cli[cli.index("-t") + 1]) if "-t" in cli else 0.0
    return Scanner(case["cap"]["sr"], fq=[fq] * n_channels, iq_mode=case["mode"], dc=case["dc"], bw_khz=case["bw"],
                   cont="-c" in cli, time_limit=tl, **kw)


def _feed(sc, x, chunk, per):
    D = sc.info["decM"]
    n = len(x) // per
    n -= n % D
    wins, dets = [], []
    for s0 in range(0, n, chunk):
        s1 = min(n, s0 + chunk)
        sc.process_host(x[per * s0:per * s1])
        wins += sc.last_windows()
        dets += sc.fetch(verbose=True)
        if sc.done(0):
            break
    return wins, dets


def _check_windows(wins, g, upto=None, tol_first=2e-5, tol=2e-5):
    n = len(wins) if upto is None else upto
    assert n <= len(g["pos"])
    for w in range(n):
        r = wins[w]
        assert r["pos"] == g["pos"][w]
        assert np.array_equal(r["mp"], g["mp"][w]), (w, r["mp"], g["mp"][w])
        ok = g["mp"][w] > 0
        assert np.array_equal(r["mpos"][ok], g["mpos"][w][ok]), w
        assert np.abs(r["mv"][ok] - g["mv"][w][ok]).max() < (tol_first if w == 0 else tol), (w, r["mv"], g["mv"][w])
        assert np.abs(r["dc"] - g["dc"][w]).max() < 1e-6
        assert np.array_equal(r["herrs"][ok], g["herrs"][w][ok]), (w, r["herrs"], g["herrs"][w])
        assert np.array_equal(r["m10"][ok], g["m10"][w][ok])


@pytest.mark.parametrize("name", [n for n in SCAN_NAMES if "imet" not in n])
def test_scan_windows_and_lines_match_reference(name):
    g = load_scan(name)
    x, fq, _, case = scan_capture(name)
    sr = case["cap"]["sr"]
    sc = _scanner(case, fq, max_chunk=sr, exact=True)
    assert sc.info["K"] == g["consts"]["K"] and sc.info["delay"] == g["consts"]["delay"] and sc.info["L"] == g["consts"]["L"]
    wins, dets = _feed(sc, x, sr // 2, 2 if case["mode"] else 1)
    # without -c / with -t the reference stops early; the harness behind the fixture always runs to the end.
    # --bw 96 at an IF rate of 96 kHz makes the single IF low-pass a unit impulse with taps of 1e-17 beside it (f_lp = 0.5, dft_detect.c:1131-1139):
    # its first 48 outputs — before the first sample has reached the centre tap — are sums of such products, and the discriminator takes the ANGLE
    # of that rounding noise (48 samples of +-0.8 in the filtered streams of window 0).  Their value depends on the reference's summation order
    # in ring-index order under its own compiler flags; window 0 of the filtered streams is compared at 5e-4 there, the unfiltered stream's
    # templates (3e-8 off) and every later window at the usual 2e-5, and the printed lines are identical.
    _check_windows(wins, g, tol_first=5e-4 if name == "scan_m10_2400k_bw96_dc" else 2e-5)
    if "-c" in case["cli"]:
        assert len(wins) == len(g["pos"])
    v = "-v" in case["cli"]
    text = "".join((d["line"] if v else d["line"].split("\n")[-1]) + "\n" for d in dets if d["printed"])
    assert text == g["stdout"]
    code = sc.result(0)
    assert code % 256 == g["rc"]


THRES = [0.65, 0.70, 0.70, 0.60, 0.80, 0.70, 0.76, 0.70, 0.70, 0.80, 0.65, 0.80, 0.65, 0.65, 0.80, 0.80]      # rs_hdr[].thres (dft_detect.c:172-191)
MARGIN = 0.003          # what the candidate test takes off the threshold; the signal-dependent part of the bound is inside smax


def _check_windows_prefiltered(wins, g):
    """default mode: pairs the exact kernel saw (herrs != -2) equal the reference; pairs left to the prefilter are below the threshold by
    more than half the margin in the REFERENCE's own values, and the prefilter's score is close to the reference's"""
    n_pre = n_exact = 0
    for w, r in enumerate(wins):
        assert r["pos"] == g["pos"][w]
        for j in range(16):
            gm = g["mp"][w][j]
            if r["herrs"][j] == -2:
                n_pre += 1
                if gm > 0:
                    assert abs(g["mv"][w][j]) < THRES[j] - MARGIN / 2, (w, j, g["mv"][w][j])
                    if r["mp"][j] == gm:
                        assert abs(r["mv"][j] - g["mv"][w][j]) < 2e-3, (w, j, r["mv"][j], g["mv"][w][j])
                continue
            if gm == 0 and r["mp"][j] == 0:
                continue
            n_exact += 1
            assert r["mp"][j] == gm, (w, j)
            assert abs(r["dc"][j] - g["dc"][w][j]) < 1e-6
            if gm > 0:
                assert r["mpos"][j] == g["mpos"][w][j] and abs(r["mv"][j] - g["mv"][w][j]) < 2e-5
                assert r["herrs"][j] == g["herrs"][w][j] and r["m10"][j] == g["m10"][w][j]
    return n_pre, n_exact


BIG = ("scan_rs41_96k_iq_dc", "scan_m10_2400k_bw96_dc", "scan_dfm_192k_iq")      # N_DFT 16384 / 32768: every pair takes the exact kernel (no prefilter)


@pytest.mark.parametrize("name", [n for n in SCAN_NAMES if "imet" not in n and n not in BIG])
def test_scan_prefilter_changes_nothing_that_is_printed(name):
    """The default mode on the captures of the per-window tests: same text lines and exit code as the reference; the exact kernel ran for a
    small part of the pairs only, and for every hit and its predecessor window."""
    g = load_scan(name)
    x, fq, _, case = scan_capture(name)
    sr = case["cap"]["sr"]
    sc = _scanner(case, fq, max_chunk=sr)
    wins, dets = _feed(sc, x, sr // 2, 2 if case["mode"] else 1)
    n_pre, n_exact = _check_windows_prefiltered(wins, g)
    # the point of it.  (Round 6: the candidate test carries the derived rounding bound, and a span without energy — the zeros in front of a stream's first sample — is
    # never ruled out by a rounding argument: the first window of a channel goes through the exact kernel with all its templates, once.  Not counted here.)
    assert n_pre > 4 * (n_exact - 14)
    v = "-v" in case["cli"]
    text = "".join((d["line"] if v else d["line"].split("\n")[-1]) + "\n" for d in dets if d["printed"])
    assert text == g["stdout"]
    assert sc.result(0) % 256 == g["rc"]
    for d in dets:                                           # every detection came from a pair the exact kernel evaluated (header compared: herrs >= 0)
        assert any(w["mpos"][d["tpl"]] == d["sample"] and w["herrs"][d["tpl"]] >= 0 for w in wins)
    # (the window before a hit is re-evaluated exactly as well; when it belongs to the previous call its record has already been handed out, so
    # that is checked through what it decides: the text lines above and test_scan_prefilter_chunk_edges)


def test_scan_prefilter_chunk_edges():
    """Windows whose predecessor belongs to the previous call: odd chunk sizes put call edges between a header's two windows"""
    name = "scan_rs41_2400k_dc"
    g = load_scan(name)
    x, fq, _, case = scan_capture(name)
    for chunk in (350 * 50 * 7, 2_400_000 // 5 + 50 * 13):
        sc = _scanner(case, fq, max_chunk=2_400_000)
        wins, dets = _feed(sc, x, chunk, 2)
        _check_windows_prefiltered(wins, g)
        assert len(wins) == len(g["pos"])
        assert "".join(d["line"] + "\n" for d in dets) == g["stdout"]


def test_scan_time_limit_stops_windows():
    g = load_scan("scan_none_48k_t2")
    x, fq, _, case = scan_capture("scan_none_48k_t2")
    sc = _scanner(case, fq, max_chunk=48000)
    wins, dets = _feed(sc, x, 48000, 2)
    # -t 2: windows while sample_in <= (2+1)*48000 (dft_detect.c:1485) -> floor(144000 / (K-4)) of them
    assert len(wins) == 144000 // (sc.info["K"] - 4)
    assert sc.done(0) and not dets and sc.result(0) == 0


def test_scan_chunking_invariance():
    """Odd chunk sizes (IQ-DC segments and windows straddle calls) give the same windows as one-second calls."""
    name = "scan_rs41_2400k_dc"
    g = load_scan(name)
    x, fq, _, case = scan_capture(name)
    sc = _scanner(case, fq, max_chunk=2_400_000, exact=True)
    D = sc.info["decM"]
    wins, dets = _feed(sc, x, 77 * 1000 * D // D * 1 + 350 * D, 2)
    _check_windows(wins, g)
    assert len(wins) == len(g["pos"])
    assert "".join(d["line"] + "\n" for d in dets) == g["stdout"]


def test_scan_multichannel_batch():
    """Three 48 kHz channels in one engine: RS41 (inverted), M10 and noise keep their single-channel results."""
    from radiosonde_auto_rx_amd.scan import Scanner, IFIQ
    names = ["scan_rs41_48k_inv", "scan_m10_48k", "scan_none_48k_t2"]
    caps = [scan_capture(n)[0] for n in names]
    n = min(len(c) for c in caps)
    X = np.stack([c[:n] for c in caps])
    sc = Scanner(48000, n_channels=3, iq_mode=IFIQ, dc=True, cont=True, max_chunk=48000)     # three default IF filters
    res = {0: [], 1: [], 2: []}
    for s0 in range(0, n // 2, 48000):
        s1 = min(n // 2, s0 + 48000)
        sc.process_host(X[:, 2 * s0:2 * s1])
        for d in sc.fetch(verbose=True):
            res[d["channel"]].append(d)
    assert [d["type"] for d in res[1]] == ["M10", "M10"] and not res[2]
    assert "".join(d["line"] + "\n" for d in res[1]) == load_scan("scan_m10_48k")["stdout"]
    assert res[0] and all(d["type"] == "RS41" and d["score"] < 0 for d in res[0])
    assert sc.result(0) % 256 == 253 and sc.result(1) == 5 and sc.result(2) == 0


@pytest.mark.parametrize("name", SCAN_NAMES)
def test_cli_dft_detect_matches_reference(name):
    """host/bin/dft_detect (C over the C ABI): stdout and exit code of the reference binary (golden)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    g = load_scan(name)
    x, fq, stdin, case = scan_capture(name)
    r = subprocess.run([os.path.join(ROOT, "host", "bin", "dft_detect")] + make_golden.scan_cli_args(case, fq), input=stdin,
                       capture_output=True, timeout=120)
    assert r.stdout.decode() == g["stdout"], r.stderr
    assert r.returncode == g["rc"]
    if case["mode"] == 5:
        assert r.stderr.decode().splitlines()[:2] == ["IF: %d" % g["consts"]["sr_if"], "dec: %d" % g["consts"]["decM"]]


def test_scan_wideband_shared_stream_10msps():
    """BASELINE config 3 in small: one 10 Msps stream, five channels mixed out of it by one engine (channel stride 0,
    decimation 200 -> the wide decimator variant), each compared with `dft_detect --IQ fq --dc` of the reference."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    from radiosonde_auto_rx_amd.scan import Scanner, BBIQ
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "scan_wide_10M.npz"), allow_pickle=False))
    x, fqs = make_golden.wide_capture()
    assert np.allclose(fqs, g["fqs"])
    sr = make_golden.WIDE_CASE["sr"]
    sc = Scanner(sr, fq=fqs, iq_mode=BBIQ, dc=True, cont=True, max_chunk=2_000_000, exact=True)
    consts = json.loads(str(g["consts"]))
    assert sc.info["decM"] == consts["decM"] == 200 and sc.info["K"] == consts["K"] and sc.info["L"] == consts["L"]
    wins = {c: [] for c in range(len(fqs))}
    dets = {c: [] for c in range(len(fqs))}
    n = len(x) // 2
    for s0 in range(0, n, 2_000_000):
        sc.process_host(x[2 * s0:2 * min(n, s0 + 2_000_000)], shared=True)
        for w in sc.last_windows():
            wins[w["channel"]].append(w)
        for d in sc.fetch(verbose=True):
            dets[d["channel"]].append(d)
    for c in range(len(fqs)):
        gc = {k: g["%s%d" % (k, c)] for k in ("mv", "mpos", "mp", "dc", "herrs", "m10", "pos")}
        assert len(wins[c]) == len(gc["pos"])
        _check_windows(wins[c], gc)
        assert "".join(d["line"] + "\n" for d in dets[c]) == str(g["stdout%d" % c])
        assert sc.result(c) % 256 == int(g["rc%d" % c])


def test_cli_dft_detect_batch_form():
    """`dft_detect -v -c --dc --IQ fq1,fq2,... - 10000000 16`: every listed offset of ONE stream scanned in the same launches — per channel the lines
    and the exit code the reference prints / returns when it is run once per offset (what auto_rx's detect_sonde() loop does, scan.py:541-547)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "scan_wide_10M.npz"), allow_pickle=False))
    x, fqs = make_golden.wide_capture()
    r = subprocess.run([os.path.join(ROOT, "host", "bin", "dft_detect"), "-v", "-c", "--dc", "--IQ", ",".join(repr(f) for f in fqs), "-",
                        str(make_golden.WIDE_CASE["sr"]), "16"], input=x.tobytes(), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr[-300:]
    lines = {c: "" for c in range(len(fqs))}
    codes = {}
    for l in r.stdout.decode().splitlines():
        if l.startswith("# "):
            _, c, fq, code = l.split()
            codes[int(c)] = int(code)
            assert abs(float(fq) - fqs[int(c)]) < 1e-6
        else:
            c, fq, rest = l.split(" ", 2)
            lines[int(c)] += rest + "\n"
    for c in range(len(fqs)):
        assert lines[c] == str(g["stdout%d" % c])
        assert codes[c] % 256 == int(g["rc%d" % c])


@pytest.mark.parametrize("name", [n for n in SCAN_NAMES if "imet" in n])
def test_scan_imet_afsk_check(name):
    """IMET preamble hits trigger the reference's extra second of spectrum analysis (IMET4 / IMET1RS / rejected), which
    shifts that channel's window phase; at EOF inside that second the decision is taken with the samples that exist."""
    g = load_scan(name)
    x, fq, _, case = scan_capture(name)
    sc = _scanner(case, fq, max_chunk=48000)
    dets = []
    n = len(x) // 2
    for s0 in range(0, n, 12000):
        sc.process_host(x[2 * s0:2 * min(n, s0 + 12000)])
        dets += sc.fetch(verbose=True)
        if sc.done(0):
            break
    sc.finish()
    dets += sc.fetch(verbose=True)
    assert "".join(d["line"] + "\n" for d in dets if d["printed"]) == g["stdout"]
    assert sc.result(0) % 256 == g["rc"]


@pytest.mark.parametrize("variant", ["tiny", "tiny_dc", "big_offset"])
def test_scan_prefilter_at_extreme_input_levels(variant):
    """FM audio a millionth of full scale (float32 WAV samples of 1e-6: f16 subnormals unless the prefilter scales its window), the same with --dc, and
    16-bit audio on an offset of a third of full scale without --dc: the default mode finds exactly what the exact mode finds, and what it prints for
    a detection is the exact kernel's value (the reference's score is normalised: the level does not matter to it either)"""
    x, fq, _, case = scan_capture("scan_rs41_audio")
    kw = {}
    if variant.startswith("tiny"):
        a = (x.astype(np.float32) / 32768.0 * 1e-6).astype(np.float32)
        kw["bits"] = 32
    else:
        a = np.clip(x.astype(np.float64) * 0.3 + 11000.0, -32768, 32767).astype(np.int16)
    case = dict(case, dc=(variant == "tiny_dc"))
    out = {}
    for exact in (True, False):
        sc = _scanner(case, fq, max_chunk=48000, exact=exact, **kw)
        wins, dets = _feed(sc, a, 24000, 1)
        out[exact] = ([(d["type"], d["sample"], d["line"]) for d in dets], sc.result(0), len(wins))
    assert out[True] == out[False]
    if variant != "big_offset":
        assert any(t == "RS41" for t, _, _ in out[True][0])


@pytest.mark.parametrize("two_pass", [False, True])
def test_scan_front_end_ragged_calls(two_pass, monkeypatch):
    """The base-rate front end fed in calls of every awkward length — fewer blocks than the decimator's history (Q-1 = 6), fewer than the IF filter's, an odd
    number of blocks (channel rows off the 16-byte grid: the checked-load path), one block, 64 / 65 / 127 blocks (one tile and a bit), calls that end on an
    IQ-DC window boundary (1500 blocks) and just behind one — gives the reference's windows and lines.  Both forms: one pass over the input (raw mix + block
    sums, the means folded out inside k_scan_if; the default) and two passes (SONDE_SCAN_TWO_PASS=1: window sums first, the mean off every sample)."""
    if two_pass:
        monkeypatch.setenv("SONDE_SCAN_TWO_PASS", "1")
    name = "scan_rs41_2400k_dc"
    g = load_scan(name)
    x, fq, _, case = scan_capture(name)
    sc = _scanner(case, fq, max_chunk=2_400_000, exact=True)
    D = sc.info["decM"]
    n = len(x) // 2 // D * D
    pattern = [3, 1, 2, 65, 127, 64, 1500 - 262, 1500, 1, 5, 1499, 7, 4001, 33, 12001]       # blocks per call; repeated until the capture is used up
    wins, dets, pos, k = [], [], 0, 0
    while pos < n:
        take = min(pattern[k % len(pattern)] * D, n - pos); k += 1
        sc.process_host(x[2 * pos:2 * (pos + take)])
        wins += sc.last_windows(); dets += sc.fetch(verbose=True)
        pos += take
    _check_windows(wins, g)
    assert len(wins) == len(g["pos"])
    assert "".join(d["line"] + "\n" for d in dets) == g["stdout"]


def test_scan_one_pass_front_end_with_a_moving_offset_in_the_passband(monkeypatch):
    """Where the one-pass front end's fold is actually exercised: a channel at the stream's centre (the IQ offset lands in the decimator's PASSBAND, E ~ 1) and an
    offset that drifts from window to window (every 1/32 s another mean; the Q-1 outputs behind each change mix two of them).  Calls of whole windows (every call
    starts ON a change of the mean: the blocks in front of it ran under the previous call's last mean) and of odd lengths.  The four FM streams equal those of the
    two-pass form (the mean off every sample, as the reference does it) to 2e-4 of their +-0.8 range — a wrong mean at one edge shows as 1e-1 on six samples."""
    from tools import synth
    from radiosonde_auto_rx_amd.scan import Scanner, BBIQ
    sr, D = 2_400_000, 50
    x = synth.rs41_capture(sr=sr, seconds=1.5, fq=0.0, f_offset_hz=900.0, amp=0.05, noise_sigma=0.004, t_first=0.1, seed=77).astype(np.float64)
    n = len(x) // 2 // D * D
    t = np.arange(n) / sr
    # three channels (per-channel tables, means, corrections, histories): the same signal under three different moving offsets, three carriers in the passband
    X = []
    for a_i, a_q, ph in ((0.10, -0.07, 0.0), (-0.05, 0.12, 1.0), (0.02, 0.03, 2.0)):
        y = x[:2 * n].copy()
        y[0::2] += 32768 * (a_i + 0.12 * np.sin(2 * np.pi * 1.7 * t + ph))          # I offset: moving by +-12 % of full scale at 1.7 Hz
        y[1::2] += 32768 * (a_q + 0.05 * t)                                        # Q offset: a ramp
        X.append(np.clip(np.round(y), -32768, 32767).astype(np.int16))
    X = np.stack(X)
    fqs = [0.0, synth.snap_fq(0.0004, sr), synth.snap_fq(-0.0003, sr)]
    B = sr // 32                                                                    # samples per IQ-DC window

    def run(chunks):
        sc = Scanner(sr, fq=fqs, iq_mode=BBIQ, dc=True, cont=True, max_chunk=n)
        pos, k = 0, 0
        while pos < n:
            take = min(chunks[k % len(chunks)], n - pos); k += 1
            sc.process_host(X[:, 2 * pos:2 * (pos + take)])
            pos += take
        m = n // D
        out = [sc.read_fm(c, st, m - 40000, 40000) for c in range(3) for st in range(4)]
        sc.close()
        return out

    for chunks in ([4 * B], [B, 3 * B], [7 * D * 331, 2 * B + 5 * D, 3 * D]):
        monkeypatch.delenv("SONDE_SCAN_TWO_PASS", raising=False)
        one = run(chunks)
        monkeypatch.setenv("SONDE_SCAN_TWO_PASS", "1")
        two = run(chunks)
        for st in range(12):
            assert np.abs(one[st] - two[st]).max() < 2e-4, (chunks, st, np.abs(one[st] - two[st]).max())


@pytest.mark.parametrize("two_pass", [False, True])
def test_cli_dft_detect_with_a_large_iq_offset_and_a_weak_signal_matches_the_compiled_reference(two_pass):
    """The scanner's base-rate front end under the condition its one-pass form is most exposed to: an IQ offset of 15 % / -11 % of full scale, moving,
    over a signal of 3 % of full scale.  `dft_detect -v --IQ fq --dc` on the 2.4 Msps stream: stdout (type, score to four places, offset) and the exit code of
    host/bin/dft_detect equal those of the compiled reference (oracle/_ref/dft_detect) run here on the same bytes, in the one-pass form (default) and the two-pass form."""
    from golden_cases import need_ref
    need_ref()
    from tools import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    sr = 2_400_000
    fq = synth.snap_fq(0.0003, sr)                                                   # in the decimator's passband next to DC: the offset is not filtered away
    x = synth.rs41_capture(sr=sr, seconds=2.2, fq=fq, f_offset_hz=600.0, amp=0.03, noise_sigma=0.003, t_first=0.15, seed=91).astype(np.float64)
    n = len(x) // 2
    t = np.arange(n) / sr
    x[0::2] += 32768 * (0.15 + 0.03 * np.sin(2 * np.pi * 2.3 * t))
    x[1::2] += 32768 * (-0.11 + 0.02 * t)
    raw = np.clip(np.round(x), -32768, 32767).astype(np.int16).tobytes()
    args = ["-v", "--IQ", repr(fq), "--dc", "-", str(sr), "16"]
    want = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "dft_detect")] + args, input=raw, capture_output=True, timeout=300)
    env = dict(os.environ)
    env.pop("SONDE_SCAN_TWO_PASS", None)
    if two_pass:
        env["SONDE_SCAN_TWO_PASS"] = "1"
    got = subprocess.run([os.path.join(ROOT, "host", "bin", "dft_detect")] + args, input=raw, capture_output=True, timeout=300, env=env)
    assert want.stdout.decode().strip() != "" and "RS41" in want.stdout.decode()
    assert got.stdout.decode() == want.stdout.decode(), got.stderr.decode()
    assert got.returncode == want.returncode
