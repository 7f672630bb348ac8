"""Multi-GPU sharding of the hot path: one process per GPU, clips shard with no data-path
collective, one all-gather of the fixed-width per-clip result rows at the end.

The reference is single-process/single-device (no torch.distributed anywhere); this is the
MI355X-native scale-out: ``torch.distributed`` backend ``nccl`` (= RCCL over xGMI) on the GPU
box, ``gloo`` in the CPU tests.  The payload is ~3.7 KB per clip, so the exchange is latency-
bound, not bandwidth-bound; Wav2Vec2 sequences never leave the producing GPU.
"""
from __future__ import annotations

import math


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous block of ceil(N/world) items per rank; trailing ranks may get fewer (or none)."""
    per = math.ceil(n_items / world) if world > 0 else n_items
    lo = min(n_items, rank * per)
    hi = min(n_items, lo + per)
    return lo, hi, per


def gather_rows(rows, n_total: int, group=None):
    """All-gather per-clip rows of every rank into [n_total, width] in global clip order.

    rows: [n_local, width] of this rank's shard_bounds() block.  Shards are padded to equal counts
    for the collective and the padding is trimmed afterwards."""
    import os
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return rows
    # a one-rank group has nothing to exchange; RSAF_FORCE_COLLECTIVE=1 runs the collective anyway (a 1-GPU box can then
    # show that the RCCL call executes on these buffers; bench.py creates the one-rank group for it)
    if dist.get_world_size(group) == 1 and os.environ.get("RSAF_FORCE_COLLECTIVE", "0") != "1":
        return rows
    world = dist.get_world_size(group)
    per = math.ceil(n_total / world)
    width = rows.shape[1]
    if rows.shape[0] < per:
        pad = torch.zeros((per - rows.shape[0], width), dtype=rows.dtype, device=rows.device)
        rows = torch.cat([rows, pad], dim=0)
    rows = rows.contiguous()
    out = torch.empty((world * per, width), dtype=rows.dtype, device=rows.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, rows, group=group)
    else:
        parts = [out[i * per:(i + 1) * per] for i in range(world)]
        dist.all_gather(parts, rows, group=group)
    return out[:n_total]
