"""Channel sharding across the GPUs of one node.

Channels are independent (no cross-channel state anywhere in demod_mod.c / fsk.c / dft_detect.c), so rank r of
world w simply owns a contiguous block of channels and runs the whole detect -> demod -> sync -> ECC path for
them.  The only exchange is a fixed-size per-channel detection summary per step, all-gathered over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests): <= 4096 x 16 B,
latency-bound.  No data-path collective exists.
"""
from __future__ import annotations

import numpy as np

SUMMARY_FIELDS = ("detected", "score", "pos_lo16", "ecc")   # float32 x 4 per channel


def channel_block(n_total: int, rank: int, world: int) -> range:
    """Contiguous block of global channel ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def summarize(frames: np.ndarray, n_local: int) -> np.ndarray:
    """[n_local, 4] float32 summary from a structured sonde_frame_t array (last frame of a channel wins)."""
    s = np.zeros((n_local, 4), np.float32)
    if len(frames):
        ch = frames["channel"]
        s[ch, 0] = 1.0
        s[ch, 1] = frames["mv"]
        s[ch, 2] = (frames["mv_pos"] % 65536).astype(np.float32)
        s[ch, 3] = frames["ecc"]
    return s


def gather_summaries(dist, local, world: int):
    """all_gather of equal-sized per-rank summary tensors -> list ordered by rank (global channel order)."""
    import torch
    out = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(out, local)
    return out


def max_over_ranks(dist, value: float, device) -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
