"""The device Reed-Solomon decoder behind k_framesync (csrc/sonde_rs_dev.h), on the GPU, through the C ABI:
 * word by word against the reference's bch_ecc_mod.c compiled where it lies (oracle/_ref/libref_ecc.so): 0 .. t+4 symbol errors, words
   of noise, the same rs_decode() value (0, n, -1, -2, -3) and the same bytes left in the word — repairable, unrepairable, miscorrected
   (tests/test_ecc_codes.py's matrix without its erasure columns: the hot path calls rs_decode, which has none);
 * rs41_ecc() levels 1 / 2 (2nd pass with the known block ids, both tail rules) against the pinned restatement (oracle/ora_rs41_ecc);
 * the engine: frames with bit errors are repaired in k_framesync (no host decoder call), byte-identical to the host path and the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from rs_cases import _encode, _damage, _frame, _flen, _u8

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFLIB = os.path.join(ROOT, "oracle", "_ref", "libref_ecc.so")


def _frames_from_codewords(cw1, cw2):
    """the interleaving of rs41mod.c:1730-1733 backwards: 24 + 24 parity bytes at 8 / 32, message bytes alternating from 56 on"""
    n = len(cw1)
    fr = np.zeros((n, 518), np.uint8)
    fr[:, 8:32] = cw1[:, :24]
    fr[:, 32:56] = cw2[:, :24]
    fr[:, 56:518:2] = cw1[:, 24:]
    fr[:, 57:518:2] = cw2[:, 24:]
    return fr


def _codewords_from_frames(fr):
    cw1 = np.concatenate([fr[:, 8:32], fr[:, 56:518:2]], axis=1)
    cw2 = np.concatenate([fr[:, 32:56], fr[:, 57:518:2]], axis=1)
    return cw1, cw2


@pytest.mark.skipif(not os.path.exists(REFLIB), reason="compiled reference not present (oracle/_ref)")
def test_device_decoder_word_by_word_vs_reference():
    from radiosonde_auto_rx_amd.engine import rs41_ecc_device
    ref = C.CDLL(REFLIB)
    rng = np.random.default_rng(5100)
    n = 3000
    words = np.zeros((2, n, 255), np.uint8)
    nerrs = np.zeros((2, n), int)
    for c in range(2):
        for i in range(n):
            cw = _encode(rng.integers(0, 256, 231).astype(np.uint8))
            k = int(rng.integers(0, 17)) if i % 17 else 180
            if c == 1 and i % 3 == 0:
                k = 0                                                  # a clean partner: the other wave returns at once
            nerrs[c, i] = k
            words[c, i] = _damage(cw, k, rng)
    fr = _frames_from_codewords(words[0], words[1])
    out, ecc, codes, synd = rs41_ecc_device(fr, 518, level=1)
    got = _codewords_from_frames(out)
    tally = {}
    for c in range(2):
        for i in range(n):
            b = words[c, i].copy()
            ep, ev = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
            r = ref.ref_ecc_decode(1, _u8(b), _u8(ep), _u8(ev))
            assert codes[i, c] == r, (c, i, nerrs[c, i], int(codes[i, c]), r)
            assert (got[c][i] == b).all(), (c, i, nerrs[c, i], r)
            tally[r if r < 0 else "ok"] = tally.get(r if r < 0 else "ok", 0) + 1
    e1, e2 = codes[:, 0], codes[:, 1]
    want = np.where((e1 < 0) | (e2 < 0), -((e1 < 0).astype(int) + 2 * (e2 < 0).astype(int)), e1 + e2)
    assert (ecc == want).all()
    assert tally["ok"] > 3500 and tally.get(-1, 0) > 500 and tally.get(-3, 0) + tally.get(-2, 0) >= 0, tally


def test_device_rs41_ecc_both_passes_vs_oracle():
    from oracle import bind
    from radiosonde_auto_rx_amd.engine import rs41_ecc_device
    L = bind.lib()
    rng = np.random.default_rng(5101)
    n = 1200
    frames = np.zeros((n, 518), np.uint8)
    flen = np.zeros(n, np.int32)
    for i in range(n):
        fr = _frame(rng)
        k = [0, 2, 7, 16, 22, 24, 24, 25, 26, 27, 28, 30, 36, 60][i % 14]
        fl = _flen(fr)
        pos = rng.choice(np.arange(8, fl), size=k, replace=False)
        for p in pos:
            fr[p] ^= rng.integers(1, 256)
        if i % 11 == 5:
            fr[0x38] ^= 0xF0                                           # a damaged type byte: frame length / tail rule of the other frame type
        frames[i], flen[i] = fr, _flen(fr)
    seen = set()
    for level in (1, 2):
        out, ecc, codes, synd = rs41_ecc_device(frames, flen, level=level)
        for i in range(n):
            o = np.zeros(520, np.uint8)
            o[:518] = frames[i]
            r = L.ora_rs41_ecc(_u8(o), int(flen[i]), level)
            assert ecc[i] == r, (level, i, int(ecc[i]), r)
            assert (out[i] == o[:518]).all(), (level, i)
            seen.add((level, "ok" if r >= 0 else r))
        if level == 1:
            first = ecc.copy()
        else:
            assert ((first < 0) & (ecc >= 0)).sum() > 20               # frames only the 2nd pass repairs
        # first-pass syndromes: zero exactly for the undamaged frames
        clean = np.array([i % 14 == 0 and i % 11 != 5 for i in range(n)])
        assert ((synd == 0).all(axis=1) == clean).all()
    assert {(1, "ok"), (1, -1), (1, -2), (1, -3), (2, "ok"), (2, -3)} <= seen, seen


def test_engine_repairs_frames_on_the_device(monkeypatch):
    from oracle import bind
    from tools import synth
    from radiosonde_auto_rx_amd.engine import Engine
    sr = 480_000
    specs = [(0, 11), (6, 12), (17, 13), (23, 14), (25, 15), (28, 16), (44, 17), (24, 18)]
    fqs = [synth.snap_fq(0.05 * (k - 3.5), sr) for k in range(len(specs))]
    caps = [synth.rs41_capture(sr=sr, seconds=3.3, fq=fq, seed=sd, noise_sigma=0.03, bit_errors=be) for (be, sd), fq in zip(specs, fqs)]
    n = min(len(c) for c in caps) // 2

    def run():
        eng = Engine(fqs, sr, ecc=2, max_chunk=sr)
        x = np.stack([c[:2 * n] for c in caps])
        out = []
        for pos in range(0, n, sr):
            take = min(sr, n - pos) // 10 * 10
            eng.process_host(np.ascontiguousarray(x[:, 2 * pos:2 * (pos + take)]))
            out += eng.fetch_frames()
        host = eng.host_ecc_frames()
        eng.close()
        return out, host

    dev_frames, dev_host = run()
    monkeypatch.setenv("SONDE_HOST_ECC", "1")
    host_frames, host_host = run()
    monkeypatch.delenv("SONDE_HOST_ECC")
    assert dev_host == 0 and host_host > 0                              # no whole frame went through the host decoder / the switch works
    key = lambda f: (f["channel"], f["mv_pos"])
    dev_frames.sort(key=key); host_frames.sort(key=key)
    assert [f["line"] for f in dev_frames] == [f["line"] for f in host_frames]
    assert [f["ecc"] for f in dev_frames] == [f["ecc"] for f in host_frames]
    eccs = set()
    for c, (cap, fq) in enumerate(zip(caps, fqs)):
        o = bind.ora_rs41_decode(cap[:2 * n], sr, fq=fq)
        mine = [f for f in dev_frames if f["channel"] == c]
        assert len(mine) >= 3
        for i, f in enumerate(mine):
            assert f["line"] == o["lines"][i], (c, i)
            assert f["ecc"] == int(o["ecc"][i])
            eccs.add("ok" if f["ecc"] > 0 else int(f["ecc"]))
    assert "ok" in eccs and any(isinstance(e, int) and e < 0 for e in eccs), eccs
