"""Native `imet54mod` (host/imet54mod.c: the engine's generic sonde description + include/sonde_imet54.h) on samples: stdout against the compiled
reference decoder on the same captures — IQ at 48 kHz as auto_rx feeds it (decode.py:632), SDR-rate IQ, IF-rate IQ with the centre window,
polarity (-i / --auto / skipped), FM audio."""
import os
import subprocess
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "imet54mod")
BIN = os.path.join(ROOT, "host", "bin", "imet54mod")


def _both(args, data=None):
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([BIN] + args, input=data, capture_output=True, timeout=300, env=env)
    b = subprocess.run([REF] + args, input=data, capture_output=True, timeout=300)
    assert a.returncode == b.returncode == 0, (args, a.stderr[-400:], b.stderr[-400:])
    assert a.stdout == b.stdout, (args, a.stdout[:800], b.stdout[:800])
    return a.stdout


@pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference not present")
def test_native_imet54_on_samples(tmp_path):
    from tools import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    x = synth.imet54_capture(sr=48_000, seconds=5.5, noise_sigma=0.05, seed=61)
    out = _both(["--ecc", "--IQ", "0.0", "--lpIQ", "-", "48000", "16", "--json", "--ptu"], x.tobytes())
    assert out.count(b'"type": "IMET5"') >= 4 and out.count(b"[OK]") >= 4
    y = synth.imet54_capture(sr=48_000, seconds=5.5, noise_sigma=0.15, seed=62, check="cont")
    _both(["--ecc", "-v", "--ptu", "--IQ", "0.0", "--lpIQ", "--dc", "-", "48000", "16"], y.tobytes())
    _both(["-r", "--ecc", "--iq3", "--lpIQ", "-", "48000", "16"], y.tobytes())
    _both(["-r", "--iq2", "-d", "1", "--br", "4799", "-", "48000", "16"], y.tobytes())
    _both(["-r4", "--ecc", "--iq0", "-", "48000", "16"], x.tobytes())
    inv = synth.imet54_capture(sr=48_000, seconds=4.5, noise_sigma=0.05, seed=63, invert=True)
    assert _both(["--ecc", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], inv.tobytes()) == b""
    assert _both(["--ecc", "--auto", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], inv.tobytes()).count(b"[OK]") >= 3
    assert _both(["--ecc", "-i", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], inv.tobytes()).count(b"[OK]") >= 3
    sr = 2_400_000
    fq = synth.snap_fq(0.21, sr)
    z = synth.imet54_capture(sr=sr, seconds=3.5, fq=fq, noise_sigma=0.05, seed=64)
    assert _both(["--ecc", "--json", "--jsn_cfq", "402000000", "--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"], z.tobytes()).count(b"[OK]") >= 2
    q = synth.imet54_capture(sr=48_000, seconds=5.5, noise_sigma=0.005, seed=65).astype(np.float64)
    c = q[0::2] + 1j * q[1::2]
    fm = np.angle(c[1:] * np.conj(c[:-1])) / np.pi
    pcm = np.clip(np.round(fm * 80000), -32768, 32767).astype(np.int16)
    p = tmp_path / "imet54.wav"
    with wave.open(str(p), "wb") as wv:
        wv.setnchannels(1); wv.setsampwidth(2); wv.setframerate(48000); wv.writeframes(pcm.tobytes())
    assert _both(["--ecc", "--ptu", str(p)]).count(b"[OK]") >= 4
