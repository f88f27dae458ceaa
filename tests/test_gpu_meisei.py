"""Native `meisei100mod` (host/meisei100mod.c: the engine's generic sonde description + include/sonde_meisei.h) on samples: stdout against the
compiled reference decoder on the same captures — the auto_rx form (decode.py:756: IQ at 48 kHz, --lpIQ --dc, JSON), SDR-rate IQ, IF-rate IQ,
FM audio from a file (the form that starts as iMS-100), both variants."""
import io
import os
import subprocess
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "meisei100mod")
BIN = os.path.join(ROOT, "host", "bin", "meisei100mod")


def _both(args, data=None):
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([BIN] + args, input=data, capture_output=True, timeout=300, env=env)
    b = subprocess.run([REF] + args, input=data, capture_output=True, timeout=300)
    assert a.returncode == b.returncode == 0, (args, a.stderr[-400:], b.stderr[-400:])
    assert a.stdout == b.stdout, (args, a.stdout[:800], b.stdout[:800])
    return a.stdout


@pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference not present")
def test_native_meisei_on_samples(tmp_path):
    from tools import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    x = synth.meisei_capture(sr=48_000, seconds=8.0, noise_sigma=0.05, seed=41)
    out = _both(["--IQ", "0.0", "--lpIQ", "--dc", "-", "48000", "16", "--json", "--ptu", "--ecc"], x.tobytes())
    assert out.count(b'"type": "MEISEI"') >= 5 and out.count(b"(ok)[OK]") >= 12 and b'"subtype": "IMS100"' in out
    y = synth.meisei_capture(sr=48_000, seconds=6.0, noise_sigma=0.15, seed=42, variant="rs11g")
    out = _both(["--ecc", "-v", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], y.tobytes())
    assert out.count(b"lat: 35.1") >= 4
    _both(["-r", "--ecc", "-v", "--iq2", "--lpIQ", "-", "48000", "16"], y.tobytes())
    _both(["-r", "--iq0", "-d", "1", "--br", "2399", "-", "48000", "16"], y.tobytes())
    sr = 2_400_000
    fq = synth.snap_fq(0.07, sr)
    z = synth.meisei_capture(sr=sr, seconds=3.0, fq=fq, noise_sigma=0.05, seed=43)
    out = _both(["--ecc", "--json", "--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"], z.tobytes())
    assert out.count(b"(ok)[OK]") >= 4
    # FM audio from a file: starts as iMS-100 (a file name without -r / --rs11g), RS-11G signal -> switches by itself
    q = synth.meisei_capture(sr=48_000, seconds=6.0, noise_sigma=0.01, seed=44, variant="rs11g").astype(np.float64)
    c = q[0::2] + 1j * q[1::2]
    fm = np.angle(c[1:] * np.conj(c[:-1])) / np.pi
    pcm = np.clip(np.round(fm * 40000), -32768, 32767).astype(np.int16)
    p = tmp_path / "meisei.wav"
    with wave.open(str(p), "wb") as wv:
        wv.setnchannels(1); wv.setsampwidth(2); wv.setframerate(48000); wv.writeframes(pcm.tobytes())
    out = _both(["--ecc", "--ptu", str(p)])
    assert out.count(b"lat: 35.1") >= 4
