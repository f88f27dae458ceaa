"""GPU parity of the IF-rate IQ input forms of the demodulators: `rs41mod|dfm09mod --iq0 | --iq2 | --iq3 [--iqdc]`
(dsp.opt_iq = 1 / 2 / 3, f32read_csample, demod_mod.c:419-461,763-832) — no mixer / decimator, optional running-mean
IQ-DC removal, FM discriminator (--iq0) or two-tone correlator (--iq2/3) as the sliced stream, whole-bit slicing except
--iq3 (rs41mod.c:2920-2923, dfm09mod.c:1692-1695).

Golden = the compiled reference: CLI stdout, and header hits / soft bits recorded through its own find_header /
read_softbit functions (tools/make_golden.py IFIQ_CASES).  Tolerances: text lines, stderr, header positions exact;
header score 1e-4 (the reference's FFT correlation carries ~4e-5 of twiddle drift); soft bits <= 3x the reference's own
-Ofast-vs--O2 floor stored in the fixture (+1e-6) and < 1.5e-4 absolute RMS (whole-bit sums of 10..20 samples)."""
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
NAMES = sorted(make_golden.IFIQ_CASES)


def _rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, np.float64))))) if np.size(a) else 0.0


@pytest.mark.parametrize("name", NAMES)
def test_ifiq_frames_match_reference(name):
    from radiosonde_auto_rx_amd.engine import Engine
    case = make_golden.IFIQ_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    x, _, _ = make_golden.ifiq_capture(case)
    sr = case["cap"]["sr"]
    dfm = case["gen"] == "dfm"
    eng = Engine([0.0], sr, sonde="dfm" if dfm else "rs41", ecc=1 if dfm else 2, lp_iq=case["lp_iq"], lp_fm=case["lp_fm"],
                 iq_mode=case["mode"], iqdc=case["iqdc"], bits=case.get("bits", 16), keep_soft=True, max_chunk=sr, max_frames=16)
    assert eng.info["decM"] == 1 and eng.info["if_sr"] == sr
    lines, softs, pos, mv = [], [], [], []
    n = len(x) // 2
    step = sr // 2 + 77                                       # chunks straddle the IQ-DC segments and the frames
    for s0 in range(0, n, step):
        s1 = min(n, s0 + step)
        eng.process_host(x[2 * s0:2 * s1])
        last = s1 >= n
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
    assert [l.rstrip() for l in lines] == [l.rstrip() for l in g["lines"]]
    assert pos == [int(v) for v in g["mv_pos"]]
    assert np.abs(np.array(mv) - g["mv"]).max() < 1e-4
    assert len(softs) == len(g["soft"])
    for h, s in enumerate(softs):
        nb = int(g["nbits"][h])
        d = _rms(s[:nb] - g["soft"][h][:nb])
        assert d < 1.5e-4 and d <= 3 * float(g["floor_soft"]) + 1e-6, (h, d, float(g["floor_soft"]))


@pytest.mark.parametrize("name", NAMES)
def test_cli_ifiq_matches_reference(name):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    case = make_golden.IFIQ_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    x, binary, args = make_golden.ifiq_capture(case)
    r = subprocess.run([os.path.join(ROOT, "host", "bin", binary)] + args, input=x.tobytes(), capture_output=True, timeout=180)
    assert r.returncode == int(g["rc"]), r.stderr
    assert r.stderr.decode() == str(g["stderr"])
    assert [l.rstrip() for l in r.stdout.decode().splitlines()] == [l.rstrip() for l in g["lines"]]


def test_cli_iq_in_stereo_wav():
    """IQ pairs inside a 2-channel WAV (no `- sr bits`): header summary on stderr, then the same frames."""
    from tools import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    name = "ifiq_rs41_iq2_lpIQ"
    case = make_golden.IFIQ_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    x, binary, args = make_golden.ifiq_capture(case)
    wav = synth.wav_bytes(x, case["cap"]["sr"], nch=2)
    r = subprocess.run([os.path.join(ROOT, "host", "bin", binary)] + args[:-3], input=wav, capture_output=True, timeout=180)
    assert r.returncode == 0, r.stderr
    assert "channels   : 2" in r.stderr.decode()
    assert [l.rstrip() for l in r.stdout.decode().splitlines()] == [l.rstrip() for l in g["lines"]]
