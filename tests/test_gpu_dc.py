"""GPU parity of `--dc` in the demodulators: zero-mean correlation windows, the FM-stream fallback correlation, the header
dc and the AFC feedback of find_header (demod_mod.c:174-188,227-298,758-761,1553-1600) for every input form
(--IQ fq, --iq0/2/3, FM audio), RS41 and DFM.

The captures carry a carrier offset the decoder is not told about (hundreds of Hz to 3.2 kHz = beyond the 1 kHz lock
limit, so the acquisition filter, several 60 % AFC steps and the lock switch are exercised).  Golden = the compiled
reference: CLI stdout and header hits / soft bits through its own find_header / read_softbit functions
(tools/make_golden.py DC_CASES).  Tolerances: text lines and header positions exact; header score 2e-4 (the reference
evaluates the zero-mean window through float FFTs with drifting twiddles, here it is a time-domain identity); soft
bits <= 3x the reference's own -Ofast-vs--O2 floor (+1e-6) and < 1.5e-4 RMS."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = sorted(make_golden.DC_CASES)


def _rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, np.float64))))) if np.size(a) else 0.0


def _run(case, x, fq, chunk_of):
    from radiosonde_auto_rx_amd.engine import Engine
    sr = case["cap"]["sr"]
    dfm = case["gen"] == "dfm"
    m = case["mode"]
    eng = Engine([fq], sr, sonde="dfm" if dfm else "rs41", ecc=1 if dfm else 2, lp_iq=case["lp_iq"] and m != 0, lp_fm=(m == 5),
                 opt_dc=True, audio=(m == 0), iq_mode=m if m else 5, keep_soft=True, max_chunk=sr, max_frames=16)
    D = eng.info["decM"]
    per = 1 if m == 0 else 2
    n = len(x) // per
    step = chunk_of(sr) // D * D
    lines, softs, pos, mv = [], [], [], []
    for s0 in range(0, n - n % D, step):
        s1 = min(n - n % D, s0 + step)
        eng.process_host(x[per * s0:per * s1])
        last = s1 >= n - n % D
        if dfm:
            fr, soft = eng.fetch_dfm(with_soft=True, finish=last)
            lines += [f["line"] for f in fr]
            for f in fr:
                if f["mv_pos"] not in pos:
                    pos.append(f["mv_pos"]); mv.append(f["mv"])
            softs += list(soft)
        else:
            fr = eng.fetch_frames(with_soft=True)
            if last:
                fr += eng.fetch_frames(with_soft=True, finish=True)
            lines += [f["line"] for f in fr]; pos += [f["mv_pos"] for f in fr]; mv += [f["mv"] for f in fr]
            softs += [f["soft"] for f in fr]
    eng.close()
    return lines, softs, pos, mv


@pytest.mark.parametrize("name", NAMES)
def test_dc_frames_match_reference(name):
    case = make_golden.DC_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    x, _, _, _, fq = make_golden.dc_capture(case)
    lines, softs, pos, mv = _run(case, x, fq, lambda sr: sr // 2 + 7700)
    assert [l.rstrip() for l in lines] == [l.rstrip() for l in g["lines"]]
    assert pos == [int(v) for v in g["mv_pos"]]
    assert np.abs(np.array(mv) - g["mv"]).max() < 2e-4
    assert len(softs) == len(g["soft"])
    for h, s in enumerate(softs):
        nb = int(g["nbits"][h])
        d = _rms(s[:nb] - g["soft"][h][:nb])
        assert d < 1.5e-4 and d <= 3 * float(g["floor_soft"]) + 1e-6, (h, d, float(g["floor_soft"]))


@pytest.mark.parametrize("name", ["dc_rs41_2400k_off3200", "dc_rs41_iq2_48k_off1800"])
def test_dc_chunking_invariance(name):
    """The restart loop must give the same frames whether an AFC event falls inside a chunk or on its edge."""
    case = make_golden.DC_CASES[name]
    x, _, _, _, fq = make_golden.dc_capture(case)
    a = _run(case, x, fq, lambda sr: sr)
    b = _run(case, x, fq, lambda sr: sr // 10 + 350)
    assert a[0] == b[0] and a[2] == b[2]
    for s, t in zip(a[1], b[1]):
        assert _rms(s - t) < 1e-6


@pytest.mark.parametrize("name", NAMES)
def test_cli_dc_matches_reference(name):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    case = make_golden.DC_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    _, stdin, binary, args, _ = make_golden.dc_capture(case)
    r = subprocess.run([os.path.join(ROOT, "host", "bin", binary)] + args, input=stdin, capture_output=True, timeout=180)
    assert r.returncode == int(g["rc"]), r.stderr
    assert r.stderr.decode() == str(g["stderr"])
    assert [l.rstrip() for l in r.stdout.decode().splitlines()] == [l.rstrip() for l in g["lines"]]


def test_engine_with_dc_turns_the_fm_lowpass_on_like_every_decoder():
    """`--IQ fq` with `--dc`: every decoder of the reference sets LP_FM itself (rs41mod.c:2747, dfm09mod.c:1475, m10mod.c:1320).  An engine created with opt_dc and
    WITHOUT SONDE_LP_FM must give the reference decoder's lines — found by tests/fuzz/fuzz_chunks.py (the CLIs set the flag, a library caller may not): the
    repaired-symbol counts of noisy frames are what tells the two filters apart."""
    sys.path.insert(0, ROOT)
    from tools import synth
    from radiosonde_auto_rx_amd.engine import Engine
    ref = os.path.join(ROOT, "oracle", "_ref", "rs41mod")
    if not os.path.exists(ref):
        pytest.fail("oracle/_ref/rs41mod missing: run __graft_entry__.build() where /root/reference exists")
    sr = 480_000
    fq = synth.snap_fq(-0.23, sr)
    x = synth.rs41_capture(sr=sr, seconds=4.3, fq=fq, seed=77, noise_sigma=0.08, bit_errors=12, t_first=0.4, f_offset_hz=-1452.0)
    want = [ln.rstrip() for ln in subprocess.run([ref, "-r", "--ecc2", "--IQ", repr(fq), "--lpIQ", "--dc", "-", str(sr), "16"], input=x.tobytes(),
                                                 capture_output=True, timeout=120).stdout.decode().splitlines()]
    assert len(want) >= 3
    for lp_fm in (False, True):
        eng = Engine([fq], sr, sonde="rs41", ecc=2, lp_iq=True, lp_fm=lp_fm, opt_dc=True, max_chunk=sr, max_frames=32)
        got = []
        for s0 in range(0, len(x) // 2, sr):
            m = min(sr, len(x) // 2 - s0) // 10 * 10
            eng.process_host(x[None, 2 * s0:2 * (s0 + m)])
            got += [f["line"].rstrip() for f in eng.fetch_frames()]
        got += [f["line"].rstrip() for f in eng.fetch_frames(finish=True)]
        eng.close()
        assert got == want, (lp_fm, [g[-14:] for g in got], [w[-14:] for w in want])
