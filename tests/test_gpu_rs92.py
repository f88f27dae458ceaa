"""Native `rs92mod` (host/rs92mod.c: the engine's generic sonde description + include/sonde_rs92.h) on samples: stdout against the compiled
reference decoder on the same captures — the auto_rx form (decode.py:484: `-vx -v --crc --ecc --vel --json -e <eph>` on FM audio), IF-rate
IQ through the mixer / low-pass, the centre-window and FM-sliced forms, SDR-rate IQ, an inverted signal with -i, an RS92-NGP (h = 3.8)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "rs92mod")
BIN = os.path.join(ROOT, "host", "bin", "rs92mod")


def _both(args, data=None):
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([BIN] + args, input=data, capture_output=True, timeout=300, env=env)
    b = subprocess.run([REF] + args, input=data, capture_output=True, timeout=300)
    assert a.returncode == b.returncode == 0, (args, a.stderr[-400:], b.stderr[-400:])
    assert a.stdout == b.stdout, (args, a.stdout[:800], b.stdout[:800])
    return a.stdout


@pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference not present")
def test_native_rs92_on_samples(tmp_path):
    from tools import synth, synth_rs92 as R
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    eph = R.constellation()
    E = str(tmp_path / "brdc.nav")
    open(E, "wb").write(R.rinex_nav(eph))
    fr = R.flight(6, eph)
    A = ["-vx", "-v", "--crc", "--ecc", "--vel", "--json", "-e", E]
    x = R.rs92_capture(fr, sr=48_000, noise_sigma=0.05, seed=61)
    tail = ["--IQ", "0.0", "--lpIQ", "-", "48000", "16"]
    out = _both(A + tail, x.tobytes())
    assert out.count(b'"type": "RS92"') == 6 and out.count(b'"lat": 47.71') == 6
    y = R.rs92_capture(fr, sr=48_000, noise_sigma=0.2, seed=62)
    _both(["-v", "--vel", "--ecc2", "-e", E] + tail, y.tobytes())
    _both(["-r", "-v", "--iq3", "--lpIQ", "-", "48000", "16"], y.tobytes())
    _both(["-r", "-v", "--iq2", "-d", "1", "-", "48000", "16"], y.tobytes())
    _both(["-v", "--iq0", "-", "48000", "16"], x.tobytes())
    _both(["-r", "-v", "--IQ", "0.0", "--lpIQ", "--dc", "-", "48000", "16"], y.tobytes())
    inv = R.rs92_capture(fr, sr=48_000, noise_sigma=0.05, seed=63, invert=True)
    assert _both(["-i"] + A + tail, inv.tobytes()).count(b'"type": "RS92"') == 6
    assert _both(A + tail, inv.tobytes()) == b""                                     # the other polarity is skipped (no --auto in this decoder)
    sr = 2_400_000
    fq = synth.snap_fq(-0.08, sr)
    z = R.rs92_capture(fr[:3], sr=sr, fq=fq, noise_sigma=0.05, seed=64)
    assert _both(A + ["--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"], z.tobytes()).count(b'"type": "RS92"') == 3
    # FM audio, the form auto_rx pipes in from rtl_fm
    wav = tmp_path / "rs92.wav"
    wav.write_bytes(synth.wav_bytes(synth.fm_audio(R.rs92_capture(fr, sr=48_000, noise_sigma=0.01, seed=65)), 48_000))
    assert _both(A + ["--ptu", str(wav)]).count(b'"type": "RS92"') == 6
    # RS92-NGP: wider deviation, 32 kHz IF low-pass
    caln = R.cal_rows(seed=5, freq_khz=1680500, ngp_key=bytes(range(0x31, 0x41)))
    n = R.rs92_capture(R.flight(5, eph, cal=caln, ngp=True), sr=96_000, noise_sigma=0.05, seed=66, dev_hz=9120.0)
    assert _both(["--ngp"] + A + ["--IQ", "0.0", "--lpIQ", "-", "96000", "16"], n.tobytes()).count(b'"type": "RS92"') == 5
    r = subprocess.run([BIN, "--spike", str(wav)], capture_output=True)
    assert r.returncode == 255 and b"--spike" in r.stderr
