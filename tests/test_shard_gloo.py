"""world_size-2 gloo test of the multi-GPU plumbing bench.py uses (channel blocks + summary all_gather + max time), and — on a
GPU — the summary records themselves: written by the frame-sync kernel into a torch tensor, gathered from device memory."""
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from radiosonde_auto_rx_amd import shard


def _records(blk, n_local):
    """what the frame-sync kernel would leave for the channels of one rank (padded to n_local records)"""
    r = np.zeros(n_local, shard.SUMMARY_DTYPE)
    ids = np.array(list(blk), np.uint32)
    r["channel_id"][:len(ids)] = ids
    r["type"][:len(ids)] = 41
    r["score"][:len(ids)] = 0.9 + 0.001 * ids
    r["sample_pos"][:len(ids)] = (1 << 33) + 70000 + ids.astype(np.uint64)     # beyond 32 bits: the record carries 64
    r["frames"][:len(ids)] = 1 + ids % 3
    r["frames_clean"][:len(ids)] = ids % 2
    return r


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blk = shard.channel_block(n_total, rank, world)
    n_local = max(len(shard.channel_block(n_total, r, world)) for r in range(world))   # equal-sized tensors
    local = torch.from_numpy(_records(blk, n_local).view(np.uint8).reshape(n_local, shard.SUMMARY_BYTES).copy())
    allsum = shard.gather_summaries(dist, local, world)
    tmax = shard.max_over_ranks(dist, 1.0 + rank, torch.device("cpu"))
    per_rank = shard.gather_floats(dist, 10.0 + rank, world, torch.device("cpu"))
    got = shard.decode_summaries([allsum[r][:len(shard.channel_block(n_total, r, world))] for r in range(world)])
    q.put((rank, got, tmax, per_rank))
    dist.destroy_process_group()


def test_channel_blocks_partition():
    for n, w in ((4096, 8), (1024, 4), (7, 2), (5, 8)):
        ids = [c for r in range(w) for c in shard.channel_block(n, r, w)]
        assert ids == list(range(n))


def test_summary_allgather_gloo():
    world, n_total = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _records(range(n_total), n_total)
    for rank, got, tmax, per_rank in res:
        assert tmax == 2.0 and per_rank == [10.0, 11.0]        # max over ranks, per-rank times in rank order
        assert got.dtype == shard.SUMMARY_DTYPE and len(got) == n_total
        for f in shard.SUMMARY_DTYPE.names:
            np.testing.assert_array_equal(got[f], want[f])


def _engine_summaries(device, n_ch, base, seed0):
    """one engine on `device`: n_ch RS41 channels, 2.2 s each -> (summary tensor, frames per channel)"""
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    sr = 480_000
    fqs = [synth.snap_fq(0.02 * (k + 1), sr) for k in range(n_ch)]
    caps = np.stack([synth.rs41_capture(sr=sr, seconds=2.2, fq=f, seed=seed0 + k, noise_sigma=0.02, bit_errors=(4 if k == 1 else 0)) for k, f in enumerate(fqs)])
    eng = Engine(fqs, sr, device=device, max_chunk=sr, max_frames=64)
    buf = shard.summary_buffer(n_ch, torch.device("cuda", device))
    eng.set_summary(buf.data_ptr(), base)
    n = (caps.shape[1] // 2 // 10) * 10
    frames = []
    for pos in range(0, n, sr):
        take = min(sr, n - pos)
        eng.process_host(np.ascontiguousarray(caps[:, 2 * pos:2 * (pos + take)]))
        frames += eng.fetch_frames()
    eng.set_summary(0)
    eng.close()
    return buf, frames


@pytest.mark.gpu
def test_summary_records_written_on_device():
    buf, frames = _engine_summaries(0, 3, 100, 50)
    rec = shard.decode_summaries(buf)
    from radiosonde_auto_rx_amd import engine as E
    assert list(rec["channel_id"]) == [100, 101, 102] and set(rec["type"]) == {E.SONDE_RS41}
    for ch in range(3):
        fr = [f for f in frames if f["channel"] == ch]
        assert rec["frames"][ch] == len(fr) >= 2
        assert rec["sample_pos"][ch] == fr[-1]["mv_pos"] and abs(rec["score"][ch] - fr[-1]["mv"]) < 1e-6
        assert rec["inverted"][ch] == (1 if fr[-1]["mv"] < 0 else 0)
        assert rec["frames_clean"][ch] == sum(1 for f in fr if f["ecc"] == 0)      # channel 1 carries bit errors: host ECC needed


def _gpu_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    buf, frames = _engine_summaries(rank, 2, 2 * rank, 70 + 10 * rank)
    allsum = shard.gather_summaries(dist, buf, world)             # device tensors in, device tensors out
    torch.cuda.synchronize()
    q.put((rank, shard.decode_summaries(allsum), [(f["channel"], f["mv_pos"]) for f in frames]))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_two_engines_rccl():
    """one engine per rank, summaries all_gathered from device memory over RCCL (needs 2 visible GPUs)"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7) % 2000
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, rec, _ in res:
        assert list(rec["channel_id"]) == [0, 1, 2, 3] and all(rec["frames"] >= 2)
    last = {(2 * rank + ch): pos for rank, _, fr in res for ch, pos in fr}
    for rank, rec, _ in res:
        assert [int(p) for p in rec["sample_pos"]] == [last[c] for c in range(4)]
