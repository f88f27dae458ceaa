"""GPU parity of the FM-audio input form (WAV, dsp.opt_iq = 0 — BASELINE config 1, the reference's CPU-runnable case).

No mixer / decimator / discriminator here: audio samples (optionally FM low-passed) are the sliced stream, so header
scores, positions and soft bits come from exactly representable inputs.  Tolerances: text lines and header positions
exact; soft bits 1e-6 RMS without --lpFM (sums of <= 20 floats in double), 1e-5 with the 97-tap low-pass; score 1e-4
(the reference's FFT correlation carries ~4e-5 of twiddle drift, tests/test_gpu_parity.py)."""
import os
import subprocess

import numpy as np
import pytest
from golden_cases import AUDIO_NAMES, load, audio_capture, rms

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", AUDIO_NAMES)
def test_audio_frames_match_reference(name):
    from radiosonde_auto_rx_amd.engine import Engine
    g = load(name)
    pcm, _, case = audio_capture(name)
    sr = case["cap"]["sr"]
    dfm = case["gen"] == "dfm"
    eng = Engine([0.0], sr, sonde="dfm" if dfm else "rs41", ecc=1 if dfm else 2, lp_iq=False, lp_fm=case["lpfm"], audio=True,
                 keep_soft=True, max_chunk=sr, max_frames=16)
    assert eng.info["decM"] == 1 and eng.info["if_sr"] == sr
    lines, softs, pos, mv = [], [], [], []
    n = len(pcm)
    for s0 in range(0, n, sr // 2 + 77):
        s1 = min(n, s0 + sr // 2 + 77)
        eng.process_host(pcm[s0:s1])
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
    assert [l.rstrip() for l in lines] == [l.rstrip() for l in g["lines"]]
    assert pos == [int(v) for v in g["mv_pos"]]
    assert np.abs(np.array(mv) - g["mv"]).max() < 1e-4
    tol = 1e-5 if case["lpfm"] else 1e-6
    for h, s in enumerate(softs):
        nb = int(g["nbits"][h])
        assert rms(s[:nb] - g["soft"][h][:nb]) < tol, (h, rms(s[:nb] - g["soft"][h][:nb]))
    eng.close()


def test_cli_wav_input_matches_reference_lines():
    """host/bin/rs41mod and dfm09mod on a WAV stream (no --IQ): the reference's stdout."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    for name in AUDIO_NAMES:
        g = load(name)
        _, wav, case = audio_capture(name)
        binary, args = make_golden.audio_cli(case)
        r = subprocess.run([os.path.join(ROOT, "host", "bin", binary)] + args, input=wav, capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert [l.rstrip() for l in r.stdout.decode().splitlines()] == [l.rstrip() for l in g["lines"]], name
        assert r.stderr.decode().splitlines()[:3] == ["sample_rate: %d" % case["cap"]["sr"], "bits       : 16", "channels   : 1"]
