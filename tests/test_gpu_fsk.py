"""GPU parity of the batched 2-/4-FSK modem (include/sonde_fsk.h) against the reference's utils/fsk.c.

Golden values come from the compiled reference (tools/make_golden.py: oracle/ref_fsk_harness.c drives fsk_demod_sd the
way utils/fsk_demod.c does; its soft decisions are asserted equal to the reference CLI's stdout when the fixture is made).
Tolerances:
  nin sequence, tone estimates (quantised to FFT bins), hard decisions, frame count ............. exact
  soft decisions: the kernel keeps the reference's operation order without fused multiply-adds;
  what is left is libm (atan2f in the timing estimate) .......................................... 1e-6 of the RMS, max 1e-5 of it
  norm_rx_timing 2e-7 abs, ppm 1e-3, Eb/N0 (log10f) 5e-3 dB; smoothed spectrum Sf exact (the kernel mirrors kiss_fft's radix-4/2
  butterfly network and twiddles, so estimator decisions cannot flip on near ties)
"""
import os
import subprocess

import numpy as np
import pytest
from golden_cases import FSK_NAMES, load_fsk, fsk_capture

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _modem(case, n_channels=1, max_chunk=None):
    from radiosonde_auto_rx_amd.fsk import FskModem
    return FskModem(case["cap"]["sr"], case["Rs"], n_channels=n_channels, P=case["P"], nsym=case["nsym"], fmt=case["fmt"],
                    lower=case["lower"], upper=case["upper"], mask=case["mask"], max_chunk=max_chunk or case["cap"]["sr"])


def _feed(md, x, chunk, per, ch=0):
    n = x.shape[-1] // per
    sds, recs = [], []
    for s0 in range(0, n, chunk):
        md.process_host(x[..., per * s0:per * min(n, s0 + chunk)])
        sd, rc = md.fetch(ch)
        sds.append(sd); recs += rc
    return np.concatenate(sds), recs


def _check(sd, recs, g):
    n = len(g["nin"])
    assert len(recs) == n and sd.shape == g["sd"].shape
    assert [r["nin"] for r in recs] == g["nin"].tolist()
    assert [r["nin_next"] for r in recs] == g["nin_next"].tolist()
    assert np.array_equal(np.array([r["f_est"] for r in recs], np.float32), g["f_est"])
    rms = float(np.sqrt(np.mean(g["sd"].astype(np.float64) ** 2)))
    d = sd.astype(np.float64) - g["sd"]
    assert np.sqrt(np.mean(d ** 2)) < 1e-6 * rms and np.abs(d).max() < 1e-5 * rms
    assert np.array_equal(sd < 0, g["sd"] < 0)
    assert np.abs(np.array([r["norm_rx_timing"] for r in recs]) - g["norm_rx_timing"]).max() < 2e-7
    assert np.abs(np.array([r["ppm"] for r in recs]) - g["ppm"]).max() < 1e-3
    assert np.abs(np.array([r["EbNodB"] for r in recs]) - g["EbNodB"]).max() < 5e-3
    assert np.abs(np.array([r["snr_est"] for r in recs]) - g["snr_est"]).max() < 5e-3


@pytest.mark.parametrize("name", FSK_NAMES)
def test_fsk_frames_match_reference(name):
    g = load_fsk(name)
    x, case = fsk_capture(name)
    md = _modem(case)
    assert {k: md.info[k] for k in ("Ts", "N", "Ndft", "Nmem")} == g["consts"]
    per = 1 if case["fmt"] == 1 else 2
    sd, recs = _feed(md, x, case["cap"]["sr"], per)
    _check(sd, recs, g)
    st = md.stats(0)
    assert np.array_equal(st["Sf"], g["Sf"])                  # the kernel runs kiss_fft's own butterfly network: bit for bit
    assert st["samples"] == int(g["nin"].sum())


def test_fsk_chunking_invariance():
    """Frames straddling process calls (samples stay queued) give the same stream as one-second calls."""
    name = "fsk_rs41_48k_mask"
    g = load_fsk(name)
    x, case = fsk_capture(name)
    md = _modem(case, max_chunk=7001)
    sd, recs = _feed(md, x, 7001, 2)
    _check(sd, recs, g)


@pytest.mark.parametrize("name", ["fsk_rs41_48k_mask", "fsk_dfm_50k", "fsk_m10_48080", "fsk_rs41_48k_cu8"])
def test_fsk_random_chunks_give_the_same_frames(name):
    """Calls of random length — from a few samples (launches that cannot produce a frame) to several frames — leave the same stream as
    one-second calls: the pipelined kernel carries its oscillator phases, the last f_dc samples, the smoothed spectrum and the pending
    frame length from launch to launch, and decides per launch which frames it can finish."""
    g = load_fsk(name)
    x, case = fsk_capture(name)
    per = 1 if case["fmt"] == 1 else 2
    n = x.shape[-1] // per
    rng = np.random.default_rng(11)
    N = case["cap"]["sr"] // case["Rs"] * case["nsym"]
    md = _modem(case, max_chunk=4 * N)
    sds, recs, s0 = [], [], 0
    while s0 < n:
        c = int(rng.choice([7, 33, N // 3, N - 1, N, N + 1, 2 * N + 5, 3 * N + 17]))
        md.process_host(x[..., per * s0:per * min(n, s0 + c)])
        sd, rc = md.fetch(0)
        sds.append(sd); recs += rc
        s0 += c
    _check(np.concatenate(sds), recs, g)
    assert np.array_equal(md.stats(0)["Sf"], g["Sf"])


def test_fsk_multichannel_batch():
    """Two channels in one engine keep their single-channel results (independent nin / phase / spectrum state)."""
    ga, gb = load_fsk("fsk_rs41_48k_mask"), load_fsk("fsk_rs41_48k_cu8")
    xa, case = fsk_capture("fsk_rs41_48k_mask")
    xb = fsk_capture("fsk_rs41_48k_peak")[0]
    n = min(len(xa), len(xb))
    X = np.stack([xa[:n], xb[:n], xa[:n]])
    md = _modem(case, n_channels=3)
    sr = case["cap"]["sr"]
    out = {0: [], 2: []}
    rec = {0: [], 2: []}
    for s0 in range(0, n // 2, sr):
        md.process_host(X[:, 2 * s0:2 * min(n // 2, s0 + sr)])
        for c in (0, 2):
            sd, rc = md.fetch(c)
            out[c].append(sd); rec[c] += rc
    for c in (0, 2):
        _check(np.concatenate(out[c]), rec[c], ga)


def test_cli_fsk_demod_matches_reference_and_decodes():
    """host/bin/fsk_demod | reference rs41mod --softin: same soft decisions as the reference CLI, same decoded frames."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    for name in ("fsk_rs41_48k_mask", "fsk_dfm_50k"):
        g = load_fsk(name)
        x, case = fsk_capture(name)
        r = subprocess.run([os.path.join(ROOT, "host", "bin", "fsk_demod")] + ["--stats=5"] + make_golden.fsk_cli_args(case),
                           input=x.tobytes(), capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr[-400:]
        sd = np.frombuffer(r.stdout, np.float32).reshape(-1, case["nsym"])
        rms = float(np.sqrt(np.mean(g["sd"].astype(np.float64) ** 2)))
        assert sd.shape == g["sd"].shape and np.abs(sd - g["sd"]).max() < 1e-5 * rms and np.array_equal(sd < 0, g["sd"] < 0)
        err = r.stderr.decode().splitlines()
        assert err[0] == "Setting estimator limits to %d to %d Hz." % (case["lower"], case["upper"])
        import json
        stats = [json.loads(l) for l in err[1:]]
        assert stats and all(k in stats[0] for k in ("samples", "EbNodB", "ppm", "f1_est", "f2_est", "samp_fft"))
        assert len(stats[0]["samp_fft"]) == g["consts"]["Ndft"] // 2
        ref = os.path.join(ROOT, "oracle", "_ref", "rs41mod")
        if "rs41_lines" in g and os.path.exists(ref):
            dec = subprocess.run([ref, "--softin", "-i", "-r", "--ecc2"], input=r.stdout, capture_output=True, timeout=60)
            assert dec.stdout.decode().splitlines() == g["rs41_lines"]
        if "rs41_lines" in g:                                   # and through this repo's own soft-input framer
            dec = subprocess.run([os.path.join(ROOT, "host", "bin", "rs41mod"), "--softin", "-i", "-r", "--ecc2"], input=r.stdout,
                                 capture_output=True, timeout=60)
            assert dec.stdout.decode().splitlines() == g["rs41_lines"]
    # hard-decision output: one byte per bit
    g = load_fsk("fsk_rs41_48k_peak")
    x, case = fsk_capture("fsk_rs41_48k_peak")
    r = subprocess.run([os.path.join(ROOT, "host", "bin", "fsk_demod")] + make_golden.fsk_cli_args(case, soft=False), input=x.tobytes(),
                       capture_output=True, timeout=120)
    assert np.array_equal(np.frombuffer(r.stdout, np.uint8).reshape(-1, case["nsym"]), (g["sd"] < 0).astype(np.uint8))


def _parse_stats(text):
    """stats lines of fsk_demod --stats -> list of dicts; the eye may hold nan (not JSON): parsed leniently."""
    import json
    import re
    out = []
    for l in text.splitlines():
        if not l.startswith("{"):
            continue
        out.append(json.loads(re.sub(r"-?nan", "NaN", l)))
    return out


@pytest.mark.parametrize("name", ["fsk_rs41_48k_mask", "fsk_rs41_48k_peak", "fsk_dfm_50k", "fsk_m10_48080", "fsk_rs41_48k_cu8", "fsk_rs41_48k_real"])
def test_cli_stats_lines_match_reference(name):
    """`fsk_demod --stats=5`: every stats line of the reference (fsk_demod.c:365-411) — sample count, EbNodB, ppm, tone
    estimates, eye diagram (8 traces x 2P samples, %f) and the smoothed spectrum — at the same frames.  Tolerances: integers and
    the %.1f fields exact; eye / spectrum within 2e-6 + 1e-5 relative (printed with 6 decimals).  Lines where the reference read
    in front of its integrator array (high_sample + 1 < 0, fsk.c:869,884: garbage in trace 0, which then also scales the whole
    normalised eye) are compared without the eye."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    case = make_golden.FSK_CASES[name]
    x = make_golden.fsk_capture(case)
    ref = str(np.load(os.path.join(ROOT, "tests", "golden", name + "_stats.npz"))["stderr"])
    r = subprocess.run([os.path.join(ROOT, "host", "bin", "fsk_demod")] + ["--stats=5"] + make_golden.fsk_cli_args(case),
                       input=x.tobytes(), capture_output=True, timeout=120)
    assert r.returncode == 0
    got, want = _parse_stats(r.stderr.decode()), _parse_stats(ref)
    assert len(got) == len(want) and len(want) >= 3
    n_eye = 0
    for a, b in zip(got, want):
        assert a["samples"] == b["samples"] and a["ppm"] == b["ppm"]
        assert a["EbNodB"] == b["EbNodB"] and a["f1_est"] == b["f1_est"] and a["f2_est"] == b["f2_est"]
        fa, fb = np.array(a["samp_fft"]), np.array(b["samp_fft"])
        assert fa.shape == fb.shape and np.all(np.abs(fa - fb) <= 2e-6 + 1e-5 * np.abs(fb))
        ea, eb = np.array(a["eye_diagram"], float), np.array(b["eye_diagram"], float)
        assert ea.shape == eb.shape == (8, 2 * case["P"])
        oob = np.isnan(eb).any() or np.abs(eb[2:]).max() < 1e-3 or not np.isclose(np.nanmax(eb), 1.0)
        if not oob:
            assert np.all(np.abs(ea - eb) <= 2e-6 + 1e-5 * np.abs(eb)), (np.abs(ea - eb).max())
            n_eye += 1
    assert n_eye >= len(want) // 2


def test_fsk_channels_fed_independently_and_reset():
    """sonde_fsk_process_host_var / sonde_fsk_reset_channel (what the resident broker needs): every channel is fed exactly its own fsk_nin()
    samples per call — the counts differ between channels and calls — and a channel that is reset and fed the same stream again repeats its
    output bit for bit while its neighbour carries on."""
    ga = load_fsk("fsk_rs41_48k_mask")
    xa, case = fsk_capture("fsk_rs41_48k_mask")
    md = _modem(case, n_channels=2, max_chunk=4096)
    N = md.info["N"]

    def run(streams, frames):
        pos = [0, 0]; nin = [N, N]; out = [[], []]
        for _ in range(frames):
            chunks = []
            for c in range(2):
                x = streams[c]
                chunks.append(None if x is None or 2 * (pos[c] + nin[c]) > len(x) else x[2 * pos[c]:2 * (pos[c] + nin[c])])
            md.process_host_var(chunks)
            for c in range(2):
                if chunks[c] is None:
                    continue
                sd, recs = md.fetch(c)
                assert len(recs) == 1 and recs[0]["nin"] == nin[c]
                pos[c] += nin[c]; nin[c] = recs[0]["nin_next"]; out[c].append(sd[0])
        return [np.array(o) for o in out]

    nfr = len(ga["nin"])
    shifted = xa[2 * 777:]                                       # channel 1: the same capture 777 samples later -> other nin decisions
    first = run([xa, shifted], nfr)
    assert np.array_equal(first[0][:nfr] < 0, ga["sd"][:len(first[0])] < 0) and len(first[0]) == nfr
    rms = float(np.sqrt(np.mean(ga["sd"].astype(np.float64) ** 2)))
    assert np.abs(first[0] - ga["sd"]).max() < 1e-5 * rms
    md.reset_channel(0)
    again = run([xa, None], nfr)
    assert np.array_equal(again[0], first[0])                    # bit for bit: the reset channel starts from the fsk_create_hbr() state
    md.close()


REFBIN = os.path.join(ROOT, "oracle", "_ref", "fsk_demod")
SEAMBIN = os.path.join(ROOT, "oracle", "_ref", "fsk_demod_seam")
NATIVE = os.path.join(ROOT, "host", "bin", "fsk_demod")


def _run3(args, data, which=("native", "seam", "ref")):
    """-> {name: CompletedProcess} of this repo's CLI, the reference's fsk_demod.c on the fsk.h seam, and the all-CPU reference"""
    if not (os.path.exists(REFBIN) and os.path.exists(SEAMBIN)):
        pytest.skip("oracle/_ref fsk_demod / fsk_demod_seam not built (make -C oracle ref)")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    bins = dict(native=NATIVE, seam=SEAMBIN, ref=REFBIN)
    out = {k: subprocess.run([bins[k]] + args, input=data, capture_output=True, timeout=300) for k in which}
    for k, r in out.items():
        assert r.returncode == 0, (k, r.stderr[-400:])
    return out


def _sd_close(a: bytes, b: bytes):
    x, y = np.frombuffer(a, np.float32), np.frombuffer(b, np.float32)
    rms = float(np.sqrt(np.mean(y.astype(np.float64) ** 2)))
    assert x.shape == y.shape and len(y) > 0
    assert np.abs(x - y).max() < 1e-5 * rms and np.array_equal(x < 0, y < 0)


@pytest.mark.parametrize("name", ["fsk_rs41_48k_mask", "fsk_dfm_50k", "fsk_rs41_48k_peak"])
def test_fsk_h_seam_reference_main_on_gpu_modem(name):
    """The reference's own utils/fsk_demod.c linked against host/seam/fsk_hip.c (fsk.h:115-205 over libsonde_hip) instead of fsk.c:
    soft decisions, hard decisions and --stats lines like the all-CPU binary."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    x, case = fsk_capture(name)
    o = _run3(["--stats=5"] + make_golden.fsk_cli_args(case), x.tobytes(), which=("seam", "ref"))
    _sd_close(o["seam"].stdout, o["ref"].stdout)
    got, want = _parse_stats(o["seam"].stderr.decode()), _parse_stats(o["ref"].stderr.decode())
    assert o["seam"].stderr.decode().splitlines()[0] == o["ref"].stderr.decode().splitlines()[0]
    assert len(got) == len(want) >= 3
    for a, b in zip(got, want):
        assert a["samples"] == b["samples"] and a["ppm"] == b["ppm"] and a["EbNodB"] == b["EbNodB"]
        assert a["f1_est"] == b["f1_est"] and a["f2_est"] == b["f2_est"]
        fa, fb = np.array(a["samp_fft"]), np.array(b["samp_fft"])
        assert fa.shape == fb.shape and np.all(np.abs(fa - fb) <= 2e-6 + 1e-5 * np.abs(fb))
    o = _run3(make_golden.fsk_cli_args(case, soft=False), x.tobytes(), which=("seam", "ref"))
    assert o["seam"].stdout == o["ref"].stdout and len(o["ref"].stdout) > 0


@pytest.mark.parametrize("mask", [0, 2400])
def test_4fsk_matches_reference(mask):
    """4-FSK (fsk.c: M = 4 — four tone estimates, four down-converters / integrators, two soft bits per symbol :793-802): this repo's
    CLI and the seam against the compiled reference on a synthetic 4-tone signal; hard bits recover what was sent."""
    from tools import synth
    rng = np.random.default_rng(4)
    bits = rng.integers(0, 2, 2 * 50 * 200)
    x = synth.mfsk_capture(bits, 48000, 2400, 4, f_low=-3600.0, shift=2400.0, noise_sigma=0.12, seed=9)
    base = ["--cs16", "-p", "5", "--stats=4"] + (["--mask", str(mask)] if mask else [])
    o = _run3(base + ["-s", "4", "48000", "2400", "-", "-"], x.tobytes())
    _sd_close(o["native"].stdout, o["ref"].stdout)
    _sd_close(o["seam"].stdout, o["ref"].stdout)
    want = _parse_stats(o["ref"].stderr.decode())
    for k in ("native", "seam"):
        got = _parse_stats(o[k].stderr.decode())
        assert len(got) == len(want) >= 3
        for a, b in zip(got, want):
            assert all(a[f] == b[f] for f in ("samples", "ppm", "EbNodB", "f1_est", "f2_est", "f3_est", "f4_est"))
    h = _run3(base + ["4", "48000", "2400", "-", "-"], x.tobytes())
    assert h["native"].stdout == h["ref"].stdout == h["seam"].stdout
    rx = np.frombuffer(h["ref"].stdout, np.uint8)
    # the demodulated stream is the sent one after the modem's start-up delay
    best = max(int(np.sum(rx[d:d + 10000] == bits[:10000])) for d in range(0, 400, 2))
    assert best >= 9990


def test_testframes_ber_lines_match_reference():
    """--testframes (fsk_demod.c:239-256,319-357): the sliding 100-bit comparison against the srand(158324) frame — every `errs:` line
    of the reference, and the frames / bits / errs fields of the --stats variant."""
    from tools import synth
    bits = synth.fsk_test_frame_bits(60)
    x = synth.mfsk_capture(bits, 48000, 4800, 2, f_low=-2400.0, shift=4800.0, noise_sigma=0.4, seed=3)
    o = _run3(["--cs16", "--testframes", "2", "48000", "4800", "-", "-"], x.tobytes())
    want = o["ref"].stderr.decode().splitlines()
    assert len([l for l in want if l.startswith("errs:")]) >= 20
    assert o["native"].stderr.decode().splitlines() == want and o["seam"].stderr.decode().splitlines() == want
    o = _run3(["--cs16", "-s", "--testframes", "--stats=10", "2", "48000", "4800", "-", "-"], x.tobytes())
    want = o["ref"].stderr.decode().splitlines()[1:]
    import json
    for k in ("native", "seam"):
        got = o[k].stderr.decode().splitlines()[1:]
        assert len(got) == len(want) >= 20
        for a, b in zip(got, want):
            a, b = json.loads(a), json.loads(b)
            assert all(a[f] == b[f] for f in ("samples", "frames", "bits", "errs", "ppm", "f1_est", "f2_est")) and "eye_diagram" not in a


def test_fsk_pipeline_that_gives_up_is_repeated_frame_by_frame(monkeypatch, capfd):
    """A channel whose pipelined launch ends with frames = -1 (a wait between its waves ran out; here: the test hook SONDE_FSK_TEST_ABORT) is run again by the
    frame-at-a-time kernel from the state it had before the launch — the call succeeds, every channel's frames are the reference's, the other channels are untouched
    (a spurious expiry must not fail the whole batch)"""
    name = "fsk_rs41_48k_mask"
    g = load_fsk(name)
    x, case = fsk_capture(name)
    per = 1 if case["fmt"] == 1 else 2
    n = x.shape[-1] // per
    X = np.stack([x, x, x])
    chunk = case["cap"]["sr"] // 3 + 7

    def run():
        md = _modem(case, n_channels=3, max_chunk=chunk)
        out = [([], []) for _ in range(3)]
        for s0 in range(0, n, chunk):
            md.process_host(X[:, per * s0:per * min(n, s0 + chunk)])
            for c in range(3):
                sd, rc = md.fetch(c)
                out[c][0].append(sd); out[c][1].extend(rc)
        md.close()
        return [(np.concatenate(a), b) for a, b in out]

    plain = run()
    capfd.readouterr()
    monkeypatch.setenv("SONDE_FSK_TEST_ABORT", "1")
    again = run()
    err = capfd.readouterr().err
    assert "repeating them frame by frame" in err
    for c in range(3):
        _check(again[c][0], again[c][1], g)
        assert np.array_equal(again[c][0], plain[c][0]) and again[c][1] == plain[c][1]


LONG_FRAMES = {
    # modem frames that no CU's LDS holds (N (M + 1) samples beyond ~19000): k_fsk_demod<M, true> on a slice of global memory per workgroup — found missing by
    # tests/fuzz/fuzz_fsk.py (the CLI used to end with exit code 0 and no output)
    "2fsk_ts40_nsym300": (2, 100_000, 2500, 20, 300, 2, ["--cs16", "-s"]),
    "4fsk_ts40_nsym100_hard": (4, 192_000, 4800, 40, 100, 2, ["--cs16"]),
    "4fsk_ts16_nsym300_cu8_mask": (4, 76_800, 4800, 16, 300, 3, ["--cu8", "-s", "--mask", "4800"]),
    "2fsk_ts40_real_limits": (2, 48_000, 1200, 40, 300, 1, ["-s", "-b", "3000", "-u", "16000"]),
}


@pytest.mark.parametrize("name", sorted(LONG_FRAMES))
def test_cli_fsk_demod_frames_longer_than_lds_match_the_reference(name):
    from tools import synth
    M, Fs, Rs, P, nsym, fmt, opts = LONG_FRAMES[name]
    if not os.path.exists(REFBIN):
        pytest.fail("oracle/_ref/fsk_demod missing: run __graft_entry__.build() where /root/reference exists")
    bits = np.random.default_rng(5).integers(0, 2, 6 * nsym * (M // 2) + 40)
    x = synth.mfsk_capture(bits, Fs, Rs, M, f_low=6000.0, shift=float(Rs * (2 if name.startswith("2fsk_ts40_real") else 1)), amp=0.4, noise_sigma=0.04, seed=9)
    data = synth.to_u8(x).tobytes() if fmt == 3 else np.ascontiguousarray(x[0::2]).tobytes() if fmt == 1 else x.tobytes()
    args = opts + ["--nsym=%d" % nsym, "-p", str(P), str(M), str(Fs), str(Rs), "-", "-"]
    a = subprocess.run([NATIVE] + args, input=data, capture_output=True, timeout=300)
    b = subprocess.run([REFBIN] + args, input=data, capture_output=True, timeout=300)
    assert a.returncode == b.returncode == 0 and len(a.stdout) == len(b.stdout) > 0, (a.returncode, b.returncode, len(a.stdout), len(b.stdout), a.stderr[-200:])
    if "-s" in opts:
        fa, fb = np.frombuffer(a.stdout, np.float32), np.frombuffer(b.stdout, np.float32)
        assert np.max(np.abs(fa.astype(np.float64) - fb)) <= 2e-6 * np.sqrt(np.mean(np.square(fb.astype(np.float64))))
    else:
        assert a.stdout == b.stdout
