"""Pieces of the benchmark driver that do not need a GPU (sharding plan, duplicate check, CPU-baseline bookkeeping),
kept importable so that the CPU test-suite covers the N > 1 logic of ``bench.py`` with a stub pipeline."""
from __future__ import annotations

import os
import time

from .dist import shard_bounds

SEED_BASE = 20260000            # clip k uses PCG64(seed = 20260000 + k)  (synth.synth_clip adds it)

CONFIGS = {
    # BASELINE.json configs -> stages of the pipeline
    "e2e": ["mshds", "smile", "w2v2", "cnnlstm"],       # C5's per-GPU work (C5 itself = e2e on 8 ranks)
    "C2": ["mshds", "smile"],
    "C3": ["w2v2"],
    "C4": ["cnnlstm_only"],
}


def shard_plan(rank: int, world: int, clips_per_gpu: int, total_clips: int | None):
    """(first global clip, number of local clips, global total, scaling) of one rank.

    Weak scaling (default): every rank owns ``clips_per_gpu`` clips.  Strong scaling (``total_clips``): the contiguous
    ``shard_bounds`` block of a fixed total (BASELINE config C5: 10 000 clips over 8 ranks = 1 250 each)."""
    if total_clips is None:
        return rank * clips_per_gpu, clips_per_gpu, clips_per_gpu * world, "weak"
    lo, hi, _ = shard_bounds(total_clips, rank, world)
    return lo, hi - lo, total_clips, "strong"


def pool_members(first: int, n_local: int, pool: int):
    """Pool member (distinct synthetic clip) of every local clip: global clip g plays member g mod pool."""
    return [(first + j) % pool for j in range(n_local)]


def duplicates_bit_identical(rows, members) -> bool:
    """Size-independent property checked at full size: a clip's row must not depend on its position in the batch or on
    its batch mates, so rows of the same pool member are bit-identical (NaN == NaN)."""
    import torch
    if len(members) == 0:
        return True
    first_of = {}
    ref_idx = []
    for j, m in enumerate(members):
        first_of.setdefault(m, j)
        ref_idx.append(first_of[m])
    ref = rows[torch.as_tensor(ref_idx, dtype=torch.long, device=rows.device)]
    same = (rows == ref) | (torch.isnan(rows) & torch.isnan(ref))
    return bool(same.all().item())


def tag_rows(rows, rank: int):
    """rows [n_local, width] -> [n_local, width + 1]: the producing rank rides in a last column, so that the gathered table
    itself says which rank computed each row (the tag is bench bookkeeping, not part of the result row)."""
    import torch
    tag = torch.full((rows.shape[0], 1), float(rank), dtype=rows.dtype, device=rows.device)
    return torch.cat([rows, tag], dim=1)


def check_gathered(out_tagged, world: int, clips_per_gpu: int, total_clips, pool: int):
    """What an N-rank line has to prove about the GATHERED table (rank 0, after the timed region): every global row sits
    at its global position and was produced by the rank that owns it under the sharding plan (tag column), every rank with
    a non-empty shard contributed, and rows of the same pool member are bit-identical across the whole table, i.e. across
    GPUs.  Returns (rows without the tag column, checks dict)."""
    tags = out_tagged[:, -1].cpu().tolist()
    rows = out_tagged[:, :-1]
    want = []
    owners = 0
    for r in range(world):
        _, n_local, n_total, _ = shard_plan(r, world, clips_per_gpu, total_clips)
        want += [float(r)] * n_local
        owners += 1 if n_local > 0 else 0
    order_ok = len(want) == len(tags) and all(a == b for a, b in zip(tags, want))
    contributing = len(set(tags))
    dup_ok = duplicates_bit_identical(rows, pool_members(0, rows.shape[0], pool))
    return rows, {"gathered_rows": int(rows.shape[0]), "rows_at_their_global_position_from_their_owner_rank": bool(order_ok),
                  "ranks_contributing": int(contributing), "ranks_with_a_shard": int(owners),
                  "duplicate_clips_bit_identical_across_the_gathered_table": bool(dup_ok)}


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota and by RSAF_CPU_THREADS
    (default 16 = the CPU share of a one-GPU box; the host may show 256 CPUs it does not grant)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, int(os.environ.get("RSAF_CPU_THREADS", "16"))))


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def median_time(fn, warmup: int, repeats: int):
    """(median seconds, all samples) of ``fn()`` after ``warmup`` untimed calls."""
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts
