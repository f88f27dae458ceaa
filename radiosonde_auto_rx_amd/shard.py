"""Channel sharding across the GPUs of one node.

Channels are independent (no cross-channel state anywhere in demod_mod.c / fsk.c / dft_detect.c), so rank r of world w owns a
contiguous block of channels and runs the whole detect -> demod -> sync -> ECC path for them.  The only exchange is the fixed-size
per-channel detection summary (sonde_summary_t, 32 bytes, include/sonde_hip.h): the frame-sync kernel writes the records into a
device buffer of the caller — here a torch tensor — and one all_gather per step over torch.distributed moves them (backend "nccl" =
RCCL over xGMI on the GPU node, "gloo" in the CPU tests): <= 4096 x 32 B, latency-bound, no host round trip.  No data-path
collective exists.
"""
from __future__ import annotations

import numpy as np

SUMMARY_BYTES = 32
SUMMARY_DTYPE = np.dtype([("channel_id", "<u4"), ("type", "u1"), ("inverted", "u1"), ("reserved", "<u2"), ("score", "<f4"),
                          ("freq_offset_hz", "<f4"), ("sample_pos", "<u8"), ("frames", "<u4"), ("frames_clean", "<u4")])
assert SUMMARY_DTYPE.itemsize == SUMMARY_BYTES


def channel_block(n_total: int, rank: int, world: int) -> range:
    """Contiguous block of global channel ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def summary_buffer(n_local: int, device):
    """Zeroed [n_local, 32] uint8 tensor for Engine.set_summary(): the frame-sync kernel fills it in place."""
    import torch
    t = torch.zeros(n_local, SUMMARY_BYTES, dtype=torch.uint8, device=device)
    if t.is_cuda:
        torch.cuda.synchronize(t.device)         # the fill runs on torch's stream; the engine writes the records on its own
    return t


def decode_summaries(t) -> np.ndarray:
    """tensor(s) of records -> structured numpy array (host copy; for checks and reports, not part of the data path)"""
    import torch
    if isinstance(t, (list, tuple)):
        t = torch.cat(list(t), 0)
    return t.detach().cpu().numpy().reshape(-1, SUMMARY_BYTES).view(SUMMARY_DTYPE).reshape(-1)


def _host_staged(dist, t) -> bool:
    """gloo has no all_gather on device tensors: stage through host memory (single-GPU readiness tests; RCCL takes the device tensors)"""
    return t.is_cuda and dist.get_backend() != "nccl"


def gather_summaries(dist, local, world: int, out=None):
    """all_gather of equal-sized per-rank summary tensors (device resident) -> list ordered by rank = global channel order."""
    import torch
    if out is None:
        out = [torch.empty_like(local) for _ in range(world)]
    if _host_staged(dist, local):
        host = [torch.empty(local.shape, dtype=local.dtype) for _ in range(world)]
        dist.all_gather(host, local.cpu())
        for o, h in zip(out, host):
            o.copy_(h)
        return out
    dist.all_gather(out, local)
    return out


def max_over_ranks(dist, value: float, device) -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(dist, value: float, world: int, device) -> list:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]
