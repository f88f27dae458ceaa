"""Native `mts01mod` (host/mts01mod.c: the engine's generic sonde description + include/sonde_mts01.h) on samples: stdout against the compiled
reference decoder on the same captures — the auto_rx form (decode.py:781: `--json --IQ 0.0 --lpIQ --dc - <sr> 16`), SDR-rate IQ, IF-rate IQ
with the centre window, an inverted signal (read raw, as the reference does), FM audio."""
import os
import subprocess
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "mts01mod")
BIN = os.path.join(ROOT, "host", "bin", "mts01mod")


def _both(args, data=None):
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([BIN] + args, input=data, capture_output=True, timeout=300, env=env)
    b = subprocess.run([REF] + args, input=data, capture_output=True, timeout=300)
    assert a.returncode == b.returncode == 0, (args, a.stderr[-400:], b.stderr[-400:])
    assert a.stdout == b.stdout, (args, a.stdout[:800], b.stdout[:800])
    return a.stdout


@pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference not present")
def test_native_mts01_on_samples(tmp_path):
    from tools import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    x = synth.mts01_capture(sr=48_000, seconds=6.5, noise_sigma=0.05, seed=51)
    out = _both(["--json", "--IQ", "0.0", "--lpIQ", "--dc", "-", "48000", "16"], x.tobytes())
    assert out.count(b'"type": "MTS01"') >= 5 and out.count(b"[OK]") >= 5
    y = synth.mts01_capture(sr=48_000, seconds=6.5, noise_sigma=0.2, seed=52)
    _both(["-v", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], y.tobytes())
    _both(["-r", "--iq3", "--lpIQ", "-", "48000", "16"], y.tobytes())
    _both(["-r", "--iq2", "-d", "1", "--br", "1201", "-", "48000", "16"], y.tobytes())
    _both(["-R", "--iq0", "-", "48000", "16"], x.tobytes())
    inv = synth.mts01_capture(sr=48_000, seconds=4.5, noise_sigma=0.05, seed=53, invert=True)
    assert _both(["-r", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], inv.tobytes()).count(b"[NO]") >= 3
    sr = 2_400_000
    fq = synth.snap_fq(-0.13, sr)
    z = synth.mts01_capture(sr=sr, seconds=3.5, fq=fq, noise_sigma=0.05, seed=54)
    assert _both(["-v", "--json", "--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"], z.tobytes()).count(b"[OK]") >= 2
    q = synth.mts01_capture(sr=48_000, seconds=5.5, noise_sigma=0.005, seed=55).astype(np.float64)
    c = q[0::2] + 1j * q[1::2]
    fm = np.angle(c[1:] * np.conj(c[:-1])) / np.pi
    pcm = np.clip(np.round(fm * 200000), -32768, 32767).astype(np.int16)
    p = tmp_path / "mts01.wav"
    with wave.open(str(p), "wb") as wv:
        wv.setnchannels(1); wv.setsampwidth(2); wv.setframerate(48000); wv.writeframes(pcm.tobytes())
    assert _both(["-v", str(p)]).count(b"[OK]") >= 4
    r = subprocess.run([BIN, "--spike", str(p)], capture_output=True)
    assert r.returncode == 255 and b"--spike" in r.stderr
