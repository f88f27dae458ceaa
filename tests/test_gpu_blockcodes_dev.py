"""The block codes of the base-rate engine's DFM and M10 hits on the device (k_dfm_hits / k_m10_hits behind the frame sync, sonde_softin_dev.hip) against the
host code they replace (dfm_block / mxx_bytes + sonde_m10_frame_finish in sonde_engine_fetch_dfm / _m10, kept as the A/B path: sonde_engine_set_device_ecc(0)):
the same frames field for field — nibbles, hamming() codes per block, raw bits, checksum verdicts, positions — on clean, damaged and noisy captures, several
channels per engine, every ecc level, streams that end inside a frame.  (That either path equals the reference's decoders is what test_gpu_parity.py,
test_gpu_m10.py, test_gpu_chain.py check — with the device path, the default.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(eng, x, sr, fetch, chunk=None):
    D = eng.info["decM"]
    n = x.shape[-1] // 2
    chunk = chunk or sr
    out, soft = [], []
    for pos in range(0, n, chunk):
        take = min(chunk, n - pos) // D * D
        if take <= 0:
            break
        eng.process_host(x[..., 2 * pos:2 * (pos + take)])
        r = fetch(eng, pos + take >= n - D)
        if isinstance(r, tuple):
            out += r[0]
            hits = []
            for f in r[0]:                                  # the hits of this fetch in record order: their soft bits come in the same order
                if (f["channel"], f["mv_pos"]) not in hits:
                    hits.append((f["channel"], f["mv_pos"]))
            soft += [(h, sb.copy()) for h, sb in zip(hits, r[1])]
        else:
            out += r
    return out, soft


def _by_channel(frames):
    """(the order in which different channels' records are queued is not defined — a channel's own is)"""
    d = {}
    for f in frames:
        d.setdefault(f["channel"], []).append(f)
    return d


@pytest.mark.parametrize("ecc,sigma,errs", [(0, 0.02, 0), (1, 0.02, 1), (2, 0.02, 2), (2, 0.35, 0), (1, 0.45, 1), (2, 0.6, 3)])
def test_dfm_block_codes_on_the_device_equal_the_host_decode(ecc, sigma, errs):
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    sr = 480_000
    fqs = [synth.snap_fq(f, sr) for f in (0.1, -0.21, 0.33)]
    x = np.stack([synth.dfm_capture(sr=sr, seconds=2.6, fq=fq, noise_sigma=sigma * (1 + 0.2 * k), seed=70 + 3 * k + ecc, bit_errors_per_frame=errs, t_first=0.1 + 0.04 * k)
                  for k, fq in enumerate(fqs)])
    res = {}
    for dev in (True, False):
        eng = Engine(fqs, sr, sonde="dfm", ecc=ecc, max_chunk=sr, max_frames=64)
        eng.set_device_ecc(dev)
        res[dev] = _run(eng, x, sr, lambda e, fin: e.fetch_dfm(with_soft=True, finish=fin), chunk=sr // 2 + 30 * eng.info["decM"])
        eng.close()
    a, b = res[True], res[False]
    assert len(a[0]) == len(b[0]) and len(a[0]) >= 8
    assert _by_channel(a[0]) == _by_channel(b[0]) and sorted(_by_channel(a[0])) == [0, 1, 2]
    if errs:
        assert any(any(v != 0 for v in f["ecc"]) for f in a[0]) or ecc == 0          # (the damaged captures do exercise the decoder)
    assert len(a[1]) == len(b[1]) and sorted(h for h, _ in a[1]) == sorted(h for h, _ in b[1])
    sb_of = dict(b[1])
    for h, sa in a[1]:
        assert np.array_equal(sa, sb_of[h])                                              # the soft bits fetched afterwards (lazily, from their ring slots) are the same ones


@pytest.mark.parametrize("chk3", [False, True])
def test_m10_frames_and_checksums_on_the_device_equal_the_host_decode(chk3):
    from radiosonde_auto_rx_amd.engine import Engine, lib
    from tools import synth
    sr = 2_400_000
    fqs = [synth.snap_fq(f, sr) for f in (0.11, -0.2, 0.3, -0.05)]
    # (2.27 s: the stream ends inside channel 0's third frame)
    x = np.stack([synth.m10_capture(sr=sr, seconds=2.27, fq=fq, noise_sigma=(0.02, 0.3, 0.1, 0.5)[k], seed=120 + k, t_first=0.2 + 0.11 * k, f_offset_hz=120.0 * (k - 2),
                                    frame_fn=lambda j, k=k: synth.m10_frame(j, rng=np.random.default_rng(77 * k + j), good_checksum=(j + k) % 3 != 2))
                  for k, fq in enumerate(fqs)])
    res = {}
    for dev in (True, False):
        eng = Engine(fqs, sr, sonde="m10", max_chunk=sr, max_frames=64, keep_soft=2 if chk3 else False)
        eng.set_device_ecc(dev)
        if chk3:
            assert lib().sonde_engine_set_m10_chk3(eng._h, 1) == 0
        res[dev] = _run(eng, x, sr, lambda e, fin: e.fetch_mxx(finish=fin))[0]
        eng.close()
    a, b = res[True], res[False]
    assert len(a) == len(b) and len(a) >= 6
    assert _by_channel(a) == _by_channel(b) and sorted(_by_channel(a)) == [0, 1, 2, 3]
    assert any(f["cs_ok"] for f in a) and any(not f["cs_ok"] for f in a)
