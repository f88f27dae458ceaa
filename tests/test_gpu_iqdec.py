"""GPU parity of the front-end-only form (host/bin/iq_dec over the C ABI) against the reference's demod/mod/iq_dec.c.

Golden = stdout of the compiled reference (tools/make_golden.py).  Tolerances: float streams 1e-6 RMS / 2e-5 max
(decimated IQ is O(0.5), the reference itself is an -Ofast build); 16-bit output is the float stream x 32768
truncated (iq_dec.c:798-936), so values within float noise of an integer may land one LSB apart: |diff| <= 1 LSB
and < 1 % of the samples differ.  WAV header bytes and stderr are exact."""
import os
import subprocess

import numpy as np
import pytest
from golden_cases import IQDEC_NAMES, load_iqdec, rms

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", IQDEC_NAMES)
def test_cli_iq_dec_matches_reference(name):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    g = load_iqdec(name)
    case = make_golden.IQDEC_CASES[name]
    x, args = make_golden.iqdec_capture(case)
    r = subprocess.run([os.path.join(ROOT, "host", "bin", "iq_dec")] + args, input=x.tobytes(), capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stderr.decode() == g["stderr"]
    hdr = case.get("wav", 0)
    assert np.array_equal(np.frombuffer(r.stdout[:hdr], np.uint8), g["header"])
    out = np.frombuffer(r.stdout[hdr:], "<" + case["out"])
    assert out.shape == g["out"].shape
    if case["out"] == "f4":
        assert rms(out - g["out"]) < 1e-6 and np.abs(out - g["out"]).max() < 2e-5
    else:
        d = np.abs(out.astype(np.int32) - g["out"].astype(np.int32))
        assert d.max() <= 1 and np.mean(d != 0) < 0.01, (d.max(), np.mean(d != 0))


def test_frontend_taps_chunking_invariance():
    """Engine level: decimated-IQ / FM taps of the front-end-only mode do not depend on how the stream is chunked."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    from radiosonde_auto_rx_amd.engine import Engine, TAP_DECIM, TAP_FM
    case = make_golden.IQDEC_CASES["iqdec_2400k_bo16"]
    x, _ = make_golden.iqdec_capture(case)
    outs = []
    for chunk in (600_000, 75_000 * 3 + 50 * 7):
        eng = Engine([0.0], 2_400_000, sonde="frontend", lp_iq=False, lp_fm=True, max_chunk=600_000)
        n = len(x) // 2
        pos = 0
        while pos < n:
            take = min(chunk, n - pos) // 50 * 50
            if take <= 0:
                break
            eng.process_host(x[2 * pos:2 * (pos + take)])
            pos += take
        m = pos // 50
        outs.append((eng.read_tap(0, TAP_DECIM, 0, m), eng.read_tap(0, TAP_FM, 0, m)))
        eng.close()
    m = min(len(outs[0][1]), len(outs[1][1]))
    assert np.array_equal(outs[0][0][:m], outs[1][0][:m]) and np.array_equal(outs[0][1][:m], outs[1][1][:m])
    g = load_iqdec("iqdec_2400k_bo16")["out"].reshape(-1, 2)
    d = np.abs(np.trunc(outs[0][0][:len(g)] * 32768.0).astype(np.int32) - g)
    assert d.max() <= 1


def test_cli_iq_dec_wav_input_equals_raw_input():
    """IQ inside a 2-channel WAV (stdin or a file argument, iq_dec.c:1051-1080) gives the bytes of the same samples fed raw; the
    compiled reference agrees on the WAV form within the 1-LSB truncation noise."""
    import sys
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    from tools import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    x, _ = make_golden.iqdec_capture(make_golden.IQDEC_CASES["iqdec_2400k_bo16"])
    x = x[:2 * 600_000]
    exe = os.path.join(ROOT, "host", "bin", "iq_dec")
    raw = subprocess.run([exe, "--bo", "16", "--iq", "0.0", "-", "2400000", "16"], input=x.tobytes(), capture_output=True, timeout=120)
    wav = synth.wav_bytes(x, 2_400_000, nch=2)
    a = subprocess.run([exe, "--bo", "16", "--iq", "0.0"], input=wav, capture_output=True, timeout=120)
    assert raw.returncode == 0 and a.returncode == 0 and a.stdout == raw.stdout and len(a.stdout) > 40_000
    with tempfile.NamedTemporaryFile(suffix=".wav") as f:
        f.write(wav); f.flush()
        b = subprocess.run([exe, "--bo", "16", "--iq", "0.0", f.name], capture_output=True, timeout=120)
    assert b.stdout == raw.stdout
    ref = os.path.join(ROOT, "oracle", "_ref", "iq_dec")
    if os.path.exists(ref):
        r = subprocess.run([ref, "--bo", "16", "--iq", "0.0"], input=wav, capture_output=True, timeout=120)
        u, v = np.frombuffer(a.stdout, "<i2").astype(np.int32), np.frombuffer(r.stdout, "<i2").astype(np.int32)
        assert u.shape == v.shape and np.abs(u - v).max() <= 1
