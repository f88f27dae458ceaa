"""GPU parity of float32 input and of --noLUT (`- <sr> 32` cf32, 32-bit float WAV) through every host CLI: rs41mod / dfm09mod (--IQ, --iq0/2
with --iqdc, FM audio), dft_detect (--IQ --dc, --iq, WAV), iq_dec.  The reference takes the floats as they are (demod_mod.c:
393,434-436,487-489; dft_detect.c:530,571-573,610-612) and keeps the IQ-DC sums in double; the device does the same with
plain (untuned) mixer / FIR kernels.  Sample values are scaled so that they are not multiples of 1/32768.
Golden = stdout / stderr / exit code of the compiled reference on the same bytes (tools/make_golden.py F32_CASES).  Text output
(frames, detections, scores to %.4f, offsets to %.1f) and exit codes exact; the iq_dec float stream 1e-6 RMS / 2e-5 max."""
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


def _rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, np.float64))))) if np.size(a) else 0.0


ALL = {**{k: (v, make_golden.f32_capture) for k, v in make_golden.F32_CASES.items()},
       **{k: (v, make_golden.nolut_capture) for k, v in make_golden.NOLUT_CASES.items()}}     # --noLUT: exact fq, absolute-index phasor


@pytest.mark.parametrize("name", sorted(ALL))
def test_cli_f32_matches_reference(name):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    case, capture = ALL[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    stdin, args = capture(case)
    r = subprocess.run([os.path.join(ROOT, "host", "bin", case["binary"])] + args, input=stdin, capture_output=True, timeout=180)
    assert r.returncode == int(g["rc"]), (r.returncode, r.stderr)
    assert r.stderr.decode() == str(g["stderr"])
    ref = g["stdout"].tobytes()
    if "out" in case:
        out, want = np.frombuffer(r.stdout, "<" + case["out"]), np.frombuffer(ref, "<" + case["out"])
        assert out.shape == want.shape
        assert _rms(out - want) < 1e-6 and np.abs(out - want).max() < 2e-5
    else:
        assert [l.rstrip() for l in r.stdout.decode().splitlines()] == [l.rstrip() for l in ref.decode().splitlines()]
        assert len(ref) > 0


def test_f32_of_int16_values_equals_s16_engine():
    """cf32 samples that are exactly b/32768 must give the streams of the tuned 16-bit path within float summation noise, the same
    frames and header positions: the plain float kernels and the packed-FMA decimator implement one filter."""
    from radiosonde_auto_rx_amd.engine import Engine, TAP_DECIM, TAP_BUFS
    from tools import synth
    sr = 2_400_000
    fqs = [synth.snap_fq(0.1, sr), synth.snap_fq(-0.2, sr)]
    s16 = np.stack([synth.rs41_capture(sr=sr, seconds=1.3, fq=fq, n_frames=1, t_first=0.05, noise_sigma=0.02, seed=120 + k, dc=0.01j) for k, fq in enumerate(fqs)])
    f32 = (s16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    res = []
    for x, bits in ((f32, 32), (s16, 16)):
        n = x.shape[1] // 2
        eng = Engine(fqs, sr, keep_soft=True, max_chunk=n, max_frames=8, bits=bits)
        pos = 0
        for take in (50 * 4001, 50 * 777, n):
            take = min(take, n - pos) // 50 * 50
            if take <= 0:
                break
            eng.process_host(np.ascontiguousarray(x[:, 2 * pos:2 * (pos + take)]))
            pos += take
        fr = sorted(eng.fetch_frames(with_soft=True, finish=True), key=lambda f: f["channel"])
        taps = [(eng.read_tap(c, TAP_DECIM, 0, pos // 50), eng.read_tap(c, TAP_BUFS, 0, pos // 50)) for c in range(2)]
        res.append((fr, taps))
        eng.close()
    (fa, ta), (fb, tb) = res
    assert len(fa) == len(fb) == 2
    for a, b in zip(fa, fb):
        assert a["line"] == b["line"] and a["mv_pos"] == b["mv_pos"] and abs(a["mv"] - b["mv"]) < 1e-5
    for (da, ba), (db, bb) in zip(ta, tb):
        assert _rms(da - db) < 2e-7 and _rms(ba - bb) < 5e-6
