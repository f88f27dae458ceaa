"""world_size-2 gloo test of the multi-GPU plumbing bench.py uses (channel blocks + summary all_gather + max time)."""
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from radiosonde_auto_rx_amd import shard
from radiosonde_auto_rx_amd.engine import Engine


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blk = shard.channel_block(n_total, rank, world)
    n_local = max(len(shard.channel_block(n_total, r, world)) for r in range(world))   # equal-sized tensors
    frames = np.zeros(len(blk), Engine.FRAME_DTYPE)
    frames["channel"] = np.arange(len(blk))
    frames["mv"] = 0.9 + 0.001 * np.array(list(blk))
    frames["mv_pos"] = 70000 + np.array(list(blk))
    frames["ecc"] = np.array(list(blk)) % 3
    local = torch.from_numpy(shard.summarize(frames, n_local))
    allsum = shard.gather_summaries(dist, local, world)
    tmax = shard.max_over_ranks(dist, 1.0 + rank, torch.device("cpu"))
    got = torch.cat([allsum[r][:len(shard.channel_block(n_total, r, world))] for r in range(world)]).numpy()
    q.put((rank, got, tmax))
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
    for rank, got, tmax in res:
        assert tmax == 2.0                                    # max over ranks
        assert got.shape == (n_total, 4)
        np.testing.assert_allclose(got[:, 1], 0.9 + 0.001 * np.arange(n_total), rtol=1e-6)
        np.testing.assert_array_equal(got[:, 2], (70000 + np.arange(n_total)) % 65536)
        np.testing.assert_array_equal(got[:, 3], np.arange(n_total) % 3)
