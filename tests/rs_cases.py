"""Shared helpers of the RS(255,231) tests (tests/test_rs_dev_emu.py on the CPU, tests/test_gpu_ecc_dev.py on the GPU): codewords from the
oracle's encoder, random symbol damage, synthetic RS41 frames as rs41_ecc() sees them."""
import ctypes as C

import numpy as np


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def _encode(msg231):
    from oracle import bind
    L = bind.lib()
    cw = np.zeros(255, np.uint8)
    cw[24:] = msg231
    assert L.ora_rs255_encode(_u8(cw)) == 0
    return cw


def _damage(cw, nerr, rng):
    a = cw.copy()
    pos = rng.choice(255, size=nerr, replace=False)
    for p in pos:
        a[p] ^= rng.integers(1, 256)
    return a


def _flen(f):
    b = int(f[0x38])
    return 320 if sum(((b >> i) & 1) - ((b >> (i + 4)) & 1) for i in range(4)) >= 0 else 518     # frametype (rs41mod.c:407-415)


def _frame(rng):
    """an RS41 frame as rs41_ecc sees it: header, 48 parity bytes, 2 x 231 interleaved message bytes (std frame: the tail is the 7611 zero block)"""
    from tools import synth
    frm = np.frombuffer(bytes(synth.rs41_frame(int(rng.integers(1, 60000)), "E%07d" % int(rng.integers(0, 9999999)),
                                               rng=np.random.default_rng(int(rng.integers(1 << 30))))), np.uint8).copy()
    out = np.zeros(518, np.uint8)
    out[:len(frm)] = frm[:518]
    return out
