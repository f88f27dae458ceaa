"""The DEVICE RS(255,231) decoder (radiosonde_auto_rx_amd/csrc/sonde_rs_dev.h: one wavefront per codeword, a coefficient per lane;
rs41_ecc() on the 1024 threads of the frame-sync workgroup) executed on the CPU under tests/emu/wave_emu.h — every thread a fiber, every
cross-lane operation a rendezvous — and compared word by word with the reference's bch_ecc_mod.c compiled where it lies
(oracle/_ref/libref_ecc.so) and with the pinned restatement of rs41_ecc (oracle/liboracle.so).  The same source is compiled by hipcc
into k_framesync; tests/test_gpu_ecc_dev.py runs it there."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from rs_cases import _u8, _encode, _damage, _flen, _frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SRC = os.path.join(ROOT, "tests", "emu", "rs_dev_emu.cpp")
EMU_SO = os.path.join(ROOT, "tests", "emu", "librs_emu.so")
REFLIB = os.path.join(ROOT, "oracle", "_ref", "libref_ecc.so")
DEPS = [EMU_SRC, os.path.join(ROOT, "tests", "emu", "wave_emu.h"), os.path.join(ROOT, "radiosonde_auto_rx_amd", "csrc", "sonde_rs_dev.h")]


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(EMU_SO) or any(os.path.getmtime(d) > os.path.getmtime(EMU_SO) for d in DEPS):
        tmp = EMU_SO + ".%d.tmp" % os.getpid()
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", tmp, EMU_SRC])
        os.replace(tmp, EMU_SO)
    return C.CDLL(EMU_SO)


@pytest.mark.skipif(not os.path.exists(REFLIB), reason="compiled reference not present")
def test_wave_decoder_matches_reference_word_by_word(emu):
    ref = C.CDLL(REFLIB)
    rng = np.random.default_rng(4100)
    stats = {}
    for trial in range(260):
        cw = _encode(rng.integers(0, 256, 231).astype(np.uint8))
        # 0 .. 12 repairable, 13 .. 16 beyond the code (failures -1 / -2 / -3 and the occasional miscorrection), a few words of noise
        nerr = int(rng.integers(0, 17)) if trial % 13 else 200
        a = _damage(cw, nerr, rng)
        b = a.copy()
        ep, ev = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
        r_ref = ref.ref_ecc_decode(1, _u8(b), _u8(ep), _u8(ev))
        r_dev = emu.emu_rs255_decode(_u8(a))
        assert r_dev == r_ref, (trial, nerr, r_dev, r_ref)
        assert (a == b).all(), (trial, nerr)
        if nerr <= 12:
            assert r_dev == nerr and (a == cw).all()
        stats[r_dev if r_dev < 0 else "ok"] = stats.get(r_dev if r_dev < 0 else "ok", 0) + 1
    assert stats["ok"] > 150 and stats.get(-1, 0) > 20, stats


@pytest.mark.skipif(not os.path.exists(REFLIB), reason="compiled reference not present")
def test_wave_decoder_low_degree_syndromes_and_single_symbols(emu):
    """words the random draw rarely produces: every single-symbol error position class, errors in the parity part only, syndromes of
    low degree (the first Euclid step already ends the loop: -3 in the reference)"""
    ref = C.CDLL(REFLIB)
    rng = np.random.default_rng(4101)
    cw0 = _encode(rng.integers(0, 256, 231).astype(np.uint8))
    words = []
    for p in (0, 1, 23, 24, 25, 127, 253, 254):
        w = cw0.copy(); w[p] ^= 0x5A; words.append(w)
    for k in (2, 6, 12, 13):
        w = cw0.copy(); w[:k] ^= 0x01; words.append(w)                 # parity bytes only
    w = np.zeros(255, np.uint8); w[0] = 1; words.append(w)              # S(x) = 1 + x + x^2 + ... : one error at position 0
    for _ in range(12):                                                # arbitrary words (not near any codeword)
        words.append(rng.integers(0, 256, 255).astype(np.uint8))
    for i, w in enumerate(words):
        a, b = w.copy(), w.copy()
        ep, ev = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
        r_ref = ref.ref_ecc_decode(1, _u8(b), _u8(ep), _u8(ev))
        assert emu.emu_rs255_decode(_u8(a)) == r_ref, i
        assert (a == b).all(), i


def _syndromes(frame, flen):
    """S_j = cw(alpha^j) of both codewords with the bytes from flen on taken as zero (what k_framesync leaves in FrameRec.synd)"""
    exp, log = np.zeros(512, np.int64), np.zeros(256, np.int64)
    x = 1
    for i in range(255):
        exp[i] = x; log[x] = i
        x <<= 1
        if x & 0x100:
            x ^= 0x11D
    exp[255:510] = exp[:255]
    f = frame.astype(np.int64).copy()
    f[flen:] = 0
    out = np.zeros(48, np.uint8)
    for c in range(2):
        cw = np.concatenate([f[8 + 24 * c:32 + 24 * c], f[56 + c:518:2]])
        for j in range(24):
            y = 0
            for n in range(254, -1, -1):
                y = (exp[log[y] + j] if y else 0) ^ int(cw[n])
            out[24 * c + j] = y
    return out


def test_workgroup_rs41_ecc_matches_oracle(emu):
    from oracle import bind
    L = bind.lib()
    rng = np.random.default_rng(4102)
    seen = set()
    for trial in range(66):
        fr = _frame(rng)
        flen = _flen(fr)
        nerr = [0, 3, 9, 20, 24, 24, 26, 27, 30, 40, 60][trial % 11]
        a = fr.copy()
        pos = rng.choice(np.arange(8, flen), size=nerr, replace=False)
        for p in pos:
            a[p] ^= rng.integers(1, 256)
        if trial % 7 == 3:
            a[0x38] ^= 0xF0                                            # a damaged type byte: the other tail rule of the 2nd pass
        rets = {}
        for level in (1, 2):
            d, o = a.copy(), np.zeros(520, np.uint8)
            o[:518] = a
            flen_a = _flen(a)
            # the product form: 256 threads with the first-pass syndromes handed in (k_framesync has them); every third case computes them
            # itself, every fifth runs as 1024 threads
            synd = _syndromes(a, flen_a)
            r_dev = emu.emu_rs41_ecc(_u8(d), flen_a, level, _u8(synd) if trial % 3 else None, 1024 if trial % 5 == 0 else 256)
            r_ora = L.ora_rs41_ecc(_u8(o), flen_a, level)
            assert r_dev == r_ora, (trial, nerr, level, r_dev, r_ora)
            assert (d == o[:518]).all(), (trial, nerr, level)
            rets[level] = r_dev
            seen.add((level, "ok" if r_dev >= 0 else r_dev))
        if rets[1] < 0 <= rets[2]:
            seen.add("rescued by the 2nd pass")
    assert (2, "ok") in seen and (1, -3) in seen and (2, -3) in seen and "rescued by the 2nd pass" in seen, seen
