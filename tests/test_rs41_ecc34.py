"""rs41mod --ecc3 / --ecc4 on soft input (CPU): sonde_rs41_dec_ecc() — byte scores, erasure candidates, bit toggling and the bytes
known from earlier frames (rs41mod.c:1703-1974, :2490-2522, :2918-2962), over the library's own RS(255,231) codec (sonde_ecc.h) —
must print what the compiled reference prints for the same float32 soft-bit stream, at noise levels where --ecc2 alone loses frames.
The sample-input variant of the same options (both soft bits per bit from the GPU engine) is tests/test_gpu_seam.py."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "host", "bin", "rs41mod")
REF = os.path.join(ROOT, "oracle", "_ref", "rs41mod")


def soft_stream(noise: float, seed: int, n: int = 14, invert: bool = False) -> bytes:
    from tools import synth
    rng = np.random.default_rng(seed)
    parts = []
    for k in range(n):
        bits = synth.rs41_onair_bits(synth.rs41_frame(100 + k), preamble_bytes=40)
        parts += [bits, np.resize(np.array([0, 1], dtype=bits.dtype), 4800 - len(bits))]       # one frame per second at 4800 Bd
    b = np.concatenate(parts).astype(np.float32) * 2 - 1
    b = b + noise * rng.standard_normal(b.size)
    return (-b if invert else b).astype("<f4").tobytes()


def _pair(args, data):
    from radiosonde_auto_rx_amd import engine
    if not os.path.exists(REF):
        pytest.skip("compiled reference not present (make -C oracle ref)")
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    a = subprocess.run([NATIVE] + args, input=data, capture_output=True, timeout=120, env=dict(os.environ, SONDE_JSN_VERSION="oracle"))
    b = subprocess.run([REF] + args, input=data, capture_output=True, timeout=120)
    assert a.returncode == b.returncode == 0, a.stderr[-300:]
    assert a.stdout == b.stdout, (args, a.stdout[:300], b.stdout[:300])
    return a.stdout


@pytest.mark.parametrize("noise,seed", [(0.38, 38), (0.41, 41), (0.44, 44), (0.46, 7)])
def test_softin_ecc3_ecc4_match_reference(noise, seed):
    sd = soft_stream(noise, seed)
    plain = _pair(["--softin", "-r", "--ecc2", "--crc"], sd)
    for ecc in ("--ecc3", "--ecc4"):
        raw = _pair(["--softin", "-r", ecc, "--crc"], sd)
        _pair(["--softin", "-v", ecc, "--crc", "--ptu"], sd)
        _pair(["--softin", "--json", ecc], sd)
        assert raw.count(b"[OK]") >= plain.count(b"[OK]")
        if noise >= 0.44:
            assert raw.count(b"[OK]") > plain.count(b"[OK]")          # the list decoding recovers frames the plain decoder loses


def test_softin_ecc3_inverted_and_truncated():
    sd = soft_stream(0.42, 5, n=6, invert=True)
    out = _pair(["--softin", "-i", "-r", "--ecc3", "--crc"], sd)
    assert out.count(b"[OK]") >= 4
    assert _pair(["--softinv", "-r", "--ecc3", "--crc"], sd) == out
    _pair(["--softin", "--auto", "-r", "--ecc4", "--crc"], sd)
    _pair(["--softin", "-i", "-r", "--ecc3", "--crc"], sd[:4 * (3 * 4800 + 1700)])           # EOF inside a frame
