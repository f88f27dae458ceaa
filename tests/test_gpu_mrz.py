"""Native `mp3h1mod` (host/mp3h1mod.c: the engine's generic sonde description + include/sonde_mrz.h) on samples: stdout against the compiled
reference decoder on the same captures — IQ at 48 kHz as auto_rx feeds it (decode.py:659), SDR-rate IQ, IF-rate IQ with the centre window,
polarity, FM audio; and a lat / lon type signal, where the reference changes the number of bits it reads per header after the first good
frame (the engine is set up again with that count)."""
import os
import subprocess
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "mp3h1mod")
BIN = os.path.join(ROOT, "host", "bin", "mp3h1mod")


def _both(args, data=None, exact=True):
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([BIN] + args, input=data, capture_output=True, timeout=300, env=env)
    b = subprocess.run([REF] + args, input=data, capture_output=True, timeout=300)
    assert a.returncode == b.returncode == 0, (args, a.stderr[-400:], b.stderr[-400:])
    if exact:
        assert a.stdout == b.stdout, (args, a.stdout[:800], b.stdout[:800])
    return a.stdout, b.stdout


@pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference not present")
def test_native_mrz_on_samples(tmp_path):
    from tools import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    x = synth.mrz_capture(sr=48_000, seconds=18.5, noise_sigma=0.05, seed=71)
    out, _ = _both(["--IQ", "0.0", "--lpIQ", "-", "48000", "16", "--json", "--ptu"], x.tobytes())
    assert out.count(b"[OK]") >= 20 and out.count(b'"type": "MRZ"') >= 2
    y = synth.mrz_capture(sr=48_000, seconds=6.5, noise_sigma=0.15, seed=72)
    _both(["-vv", "--ptu", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], y.tobytes())
    # --dc: the FM-stream fallback puts a header up to (lpFMtaps - sps + 1) / 2 samples before the window; the reference's header check then reads
    # ring slots that already hold the newest samples (demod_mod.c:268,850) — the engine reads the same samples (k_framesync)
    _both(["-vv", "--ptu", "--IQ", "0.0", "--lpIQ", "--dc", "-", "48000", "16"], x[:2 * 48_000 * 7].tobytes())
    _both(["-r", "--IQ", "0.0", "--lpIQ", "--dc", "-", "48000", "16"], y.tobytes())
    _both(["-r", "--iq3", "--lpIQ", "-", "48000", "16"], y.tobytes())
    _both(["-r", "--iq2", "-d", "1", "--br", "2399.5", "-", "48000", "16"], y.tobytes())
    _both(["-R", "--iq0", "-", "48000", "16"], y.tobytes())
    inv = synth.mrz_capture(sr=48_000, seconds=4.5, noise_sigma=0.05, seed=73, invert=True)
    assert _both(["--IQ", "0.0", "--lpIQ", "-", "48000", "16"], inv.tobytes())[0].count(b"[OK]") == 0
    assert _both(["--auto", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], inv.tobytes())[0].count(b"[OK]") >= 4
    assert _both(["-i", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], inv.tobytes())[0].count(b"[OK]") >= 4
    sr = 2_400_000
    fq = synth.snap_fq(-0.17, sr)
    z = synth.mrz_capture(sr=sr, seconds=3.5, fq=fq, noise_sigma=0.05, seed=74)
    assert _both(["--ptu", "--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"], z.tobytes())[0].count(b"[OK]") >= 3
    # lat / lon frames: 362 instead of 386 bits behind the header from the first frame whose CRC holds.  The reference goes on with its
    # filters as they are, the engine restarts them 32 bits before the end of that frame: same frames decoded; raw bytes of bad ones may differ.
    w = synth.mrz_capture(sr=48_000, seconds=8.5, noise_sigma=0.05, seed=75, latlon=True)
    a, b = _both(["--ptu", "--uniq", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], w.tobytes(), exact=False)
    oka = [l for l in a.splitlines() if b"[OK]" in l]
    okb = [l for l in b.splitlines() if b"[OK]" in l]
    assert oka == okb and len(oka) >= 6
    q = synth.mrz_capture(sr=48_000, seconds=6.5, noise_sigma=0.005, seed=76).astype(np.float64)
    c = q[0::2] + 1j * q[1::2]
    fm = np.angle(c[1:] * np.conj(c[:-1])) / np.pi
    pcm = np.clip(np.round(fm * 100000), -32768, 32767).astype(np.int16)
    p = tmp_path / "mrz.wav"
    with wave.open(str(p), "wb") as wv:
        wv.setnchannels(1); wv.setsampwidth(2); wv.setframerate(48000); wv.writeframes(pcm.tobytes())
    assert _both(["--ptu", str(p)])[0].count(b"[OK]") >= 6
