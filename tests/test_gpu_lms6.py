"""Native `lms6Xmod` (host/lms6Xmod.c: the engine's generic sonde description + include/sonde_lms6.h) on samples: stdout against the compiled
reference decoder on the same captures — IQ at SDR rate, IF-rate IQ, FM audio; LMS6, forced LMS-X, and the auto detection's change of
symbol rate in mid-stream (the function-level seam follows it the same way, tests/test_gpu_seam.py)."""
import io
import os
import subprocess
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "lms6Xmod")
BIN = os.path.join(ROOT, "host", "bin", "lms6Xmod")


def _both(args, data, exact=True):
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([BIN] + args, input=data, capture_output=True, timeout=300, env=env)
    b = subprocess.run([REF] + args, input=data, capture_output=True, timeout=300)
    assert a.returncode == b.returncode == 0, (args, a.stderr[-400:], b.stderr[-400:])
    if exact:
        assert a.stdout == b.stdout, (args, a.stdout[:800], b.stdout[:800])
    return a.stdout, b.stdout


@pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference not present")
def test_native_lms6_on_samples():
    from tools import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    sr = 2_400_000
    fq = synth.snap_fq(-0.06, sr)
    x = synth.lms6_capture(sr=sr, seconds=4.0, fq=fq, noise_sigma=0.05, seed=21)
    out, _ = _both(["--vit", "--ecc", "--json", "--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"], x.tobytes())
    assert out.count(b'"type": "LMS"') >= 3 and out.count(b"[OK]") >= 3
    y = synth.lms6_capture(sr=48_000, seconds=5.0, noise_sigma=0.12, seed=22)
    tail = ["--IQ", "0.0", "--lpIQ", "-", "48000", "16"]
    assert _both(["-r", "--ecc"] + tail, y.tobytes())[0].count(b"[OK]") >= 3
    _both(["--lms6", "--vit2", "--ecc3"] + tail, y.tobytes())
    _both(["-r", "--iq3", "--lpIQ", "-", "48000", "16"], y.tobytes())
    _both(["-r", "--iq0", "-d", "1", "-", "48000", "16"], y.tobytes())
    _both(["--lmsX", "-r"] + tail, y.tobytes())
    w = synth.lms6_capture(sr=48_000, seconds=6.0, noise_sigma=0.05, seed=31, baud=4797.8, lmsx=True)
    out, _ = _both(["--lmsX", "-r", "--ecc"] + tail, w.tobytes())
    assert out.count(b"[OK]") >= 5 and out.startswith(b"24 46 05 00")
    # auto detection on an LMS-X signal: starts as LMS6 (4800 Bd, 4096 bits), the first block shows the LMS-X frame sync, the bit clock changes to
    # 4797.8 Bd from the end of that block on.  The reference carries its filters over that point, the engine restarts them 64 bits earlier:
    # the decoded frames are the same, the raw bytes of frames with errors need not be.
    a, b = _both(["--vit", "--ecc", "--json"] + tail, w.tobytes(), exact=False)
    ja = [l for l in a.splitlines() if l.startswith(b"{")]
    jb = [l for l in b.splitlines() if l.startswith(b"{")]
    assert ja == jb and len(ja) >= 4 and b'"subtype": "LMSX-403"' in ja[0]
    # FM audio (what auto_rx pipes in from rtl_fm, decode.py:731): a discriminator output of the same signal as 16-bit mono WAV
    z = synth.lms6_capture(sr=48_000, seconds=5.0, noise_sigma=0.01, seed=23).astype(np.float64)
    c = z[0::2] + 1j * z[1::2]
    fm = np.angle(c[1:] * np.conj(c[:-1])) / np.pi
    pcm = np.clip(np.round(fm * 20000), -32768, 32767).astype(np.int16)
    buf = io.BytesIO()
    with wave.open(buf, "wb") as wv:
        wv.setnchannels(1); wv.setsampwidth(2); wv.setframerate(48000); wv.writeframes(pcm.tobytes())
    out, _ = _both(["--json"], buf.getvalue())
    assert out.count(b"[OK]") >= 3
    out, _ = _both(["--vit2", "--ecc"], buf.getvalue())            # `info: soft decoding only for IQ` -> hard decisions on both sides
    assert out.count(b"[OK]") >= 3
