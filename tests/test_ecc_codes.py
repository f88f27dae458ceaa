"""Block codes of include/sonde_ecc.h against the reference's bch_ecc_mod.c compiled where it lies (oracle/_ref/libref_ecc.so, test
infrastructure): RS(255,231), RS(255,223) CCSDS, BCH(63,51), RS(15,11) — encoder, errors-only decoder, errors-and-erasures
decoder and the binary BCH decoder, word by word on random words including those beyond the code's capability (same negative
return code, same bytes left in the word: unrepairable and miscorrected words must leave the decoder exactly alike)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFLIB = os.path.join(ROOT, "oracle", "_ref", "libref_ecc.so")
LIB = os.path.join(ROOT, "radiosonde_auto_rx_amd", "libsonde_hip.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(REFLIB) and os.path.exists(LIB)), reason="compiled reference / library not present")
CODES = {1: "RS(255,231)", 2: "RS(255,223) CCSDS", 3: "BCH(63,51)", 4: "RS(15,11)"}


def _libs():
    ref = C.CDLL(REFLIB)
    if not hasattr(ref, "ref_ecc_decode_era"):
        pytest.skip("libref_ecc.so predates the generic harness (rebuild oracle/_ref)")
    ours = C.CDLL(LIB)
    ours.sonde_ecc_create.restype = C.c_void_p
    for f in ("sonde_ecc_encode", "sonde_ecc_decode", "sonde_ecc_decode_errera", "sonde_ecc_decode_bch_gf2t2", "sonde_ecc_params", "sonde_ecc_destroy"):
        getattr(ours, f).argtypes = None
    return ref, ours


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


@pytest.mark.parametrize("code", list(CODES))
def test_code_matches_reference(code):
    ref, ours = _libs()
    h = C.c_void_p(ours.sonde_ecc_create(code))
    assert h
    N, t, R, K = (C.c_int() for _ in range(4))
    assert ours.sonde_ecc_params(h, C.byref(N), C.byref(t), C.byref(R), C.byref(K)) == 0
    n2, t2, r2, k2 = (C.c_int() for _ in range(4))
    assert ref.ref_ecc_params(code, C.byref(n2), C.byref(t2), C.byref(r2), C.byref(k2)) == 0
    assert (N.value, t.value, R.value, K.value) == (n2.value, t2.value, r2.value, k2.value)
    N, t, R = N.value, t.value, R.value
    q = 2 if code == 3 else (16 if code == 4 else 256)
    rng = np.random.default_rng(100 + code)
    stats = {"ok": 0, "fail": 0}
    for trial in range(400):
        msg = np.zeros(255, np.uint8)
        msg[R:N] = rng.integers(0, q, N - R)
        a, b = msg.copy(), msg.copy()
        assert ours.sonde_ecc_encode(h, _u8(a)) == 0 and ref.ref_ecc_encode(code, _u8(b)) == 0
        assert (a == b).all()
        # errors from 0 to t + 3 (beyond the capability in the last rounds), erasures 0 .. a few
        nerr = int(rng.integers(0, t + 4))
        nera = 0 if code == 3 else int(rng.integers(0, 5)) * int(trial % 3 == 0)
        pos = rng.choice(N, size=min(N, nerr + nera), replace=False)
        for ppos in pos[:nerr]:
            a[ppos] ^= rng.integers(1, q)
        era = np.zeros(32, np.uint8)
        era[:nera] = pos[nerr:nerr + nera]
        for ppos in pos[nerr:nerr + nera]:
            a[ppos] = rng.integers(0, q)                       # an erased position holds anything
        b = a.copy()
        ep1, ev1, ep2, ev2 = (np.zeros(64, np.uint8) for _ in range(4))
        if code == 3 and trial % 2:
            r1 = ours.sonde_ecc_decode_bch_gf2t2(h, _u8(a), _u8(ep1), _u8(ev1))
            r2_ = ref.ref_ecc_decode_bch(code, _u8(b), _u8(ep2), _u8(ev2))
        elif nera:
            r1 = ours.sonde_ecc_decode_errera(h, _u8(a), nera, _u8(era), _u8(ep1), _u8(ev1))
            r2_ = ref.ref_ecc_decode_era(code, _u8(b), nera, _u8(era), _u8(ep2), _u8(ev2))
        else:
            r1 = ours.sonde_ecc_decode(h, _u8(a), _u8(ep1), _u8(ev1))
            r2_ = ref.ref_ecc_decode(code, _u8(b), _u8(ep2), _u8(ev2))
        assert r1 == r2_, (CODES[code], trial, nerr, nera, r1, r2_)
        assert (a == b).all(), (CODES[code], trial, nerr, nera)
        if r1 > 0:
            assert (ep1[:r1] == ep2[:r1]).all() and (ev1[:r1] == ev2[:r1]).all()
        stats["ok" if r1 >= 0 else "fail"] += 1
    assert stats["ok"] > 100 and stats["fail"] > 20, stats        # both sides of the capability were exercised
    ours.sonde_ecc_destroy(h)
