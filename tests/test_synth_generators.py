"""The signal generators of tools/synth.py that have no fixture of their own are only worth something if the REFERENCE decodes what they emit:
CPU checks against the compiled reference binaries (oracle/_ref, built from the sources where they lie), no GPU involved."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def _ref(name):
    p = os.path.join(REF, name)
    if not os.path.exists(p):
        pytest.skip("compiled reference not present (make -C oracle ref)")
    return p


def test_lms6_and_lmsx_generators_decode_on_the_reference():
    from tools import synth
    x = synth.lms6_capture(sr=48_000, seconds=4.0, noise_sigma=0.05, seed=2)
    r = subprocess.run([_ref("lms6Xmod"), "--vit", "--ecc", "--json", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], input=x.tobytes(), capture_output=True, timeout=120)
    out = r.stdout.decode()
    assert out.count("[OK]") >= 3 and '"id": "LMS6-8123456"' in out and "lat: 47.50000  lon: 8.70000" in out
    y = synth.lms6_capture(sr=48_000, seconds=5.0, noise_sigma=0.05, seed=3, baud=4797.8, lmsx=True)
    r = subprocess.run([_ref("lms6Xmod"), "--lmsX", "-r", "--ecc", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], input=y.tobytes(), capture_output=True, timeout=120)
    assert r.stdout.count(b"[OK]") >= 4 and r.stdout.startswith(b"24 46 05 00")


def test_ccsds_rs_parity_matches_the_library_codec():
    """synth.rs255_223_ccsds_parity (numpy) against sonde_ecc (which tests/test_ecc_codes.py pins to the reference's bch_ecc_mod.c)"""
    import ctypes as C
    from radiosonde_auto_rx_amd import engine
    from tools import synth
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    lib = C.CDLL(engine.LIB_PATH)
    lib.sonde_ecc_create.restype = C.c_void_p
    lib.sonde_ecc_create.argtypes = [C.c_int]
    lib.sonde_ecc_encode.argtypes = [C.c_void_p, C.c_void_p]
    c = lib.sonde_ecc_create(2)
    rng = np.random.default_rng(5)
    for _ in range(5):
        msg = rng.integers(0, 256, 223, dtype=np.uint8)
        cw = np.zeros(255, np.uint8); cw[32:] = msg
        lib.sonde_ecc_encode(c, cw.ctypes.data_as(C.c_void_p))
        assert np.array_equal(synth.rs255_223_ccsds_parity(msg), cw[:32])


def test_mfsk_generator_and_testframe_bits_on_the_reference_modem():
    from tools import synth
    bits = synth.fsk_test_frame_bits(40)
    x = synth.mfsk_capture(bits, 48000, 4800, 2, f_low=-2400.0, shift=4800.0, noise_sigma=0.15, seed=3)
    r = subprocess.run([_ref("fsk_demod"), "--cs16", "--testframes", "2", "48000", "4800", "-", "-"], input=x.tobytes(), capture_output=True, timeout=120)
    lines = [l for l in r.stderr.decode().splitlines() if l.startswith("errs:")]
    assert len(lines) >= 30 and lines[-1].endswith("bit errors 0")
    rng = np.random.default_rng(4)
    b4 = rng.integers(0, 2, 2 * 50 * 60)
    y = synth.mfsk_capture(b4, 48000, 2400, 4, f_low=-3600.0, shift=2400.0, noise_sigma=0.05, seed=9)
    r = subprocess.run([_ref("fsk_demod"), "--cs16", "-p", "5", "4", "48000", "2400", "-", "-"], input=y.tobytes(), capture_output=True, timeout=120)
    rx = np.frombuffer(r.stdout, np.uint8)
    assert max(int(np.sum(rx[d:d + 4000] == b4[:4000])) for d in range(0, 400, 2)) >= 3990
