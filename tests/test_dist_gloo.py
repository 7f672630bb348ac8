"""world_size-2 gloo test of the sharding + terminal all-gather (the N>1 path of bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from robust_speech_analysis_framework_amd.dist import gather_rows, shard_bounds


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 10, 1000, 10000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi, per = shard_bounds(n, r, world)
                assert 0 <= hi - lo <= per
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert shard_bounds(10000, 7, 8) == (8750, 10000, 1250)        # BASELINE config 5: 1 250 clips / GPU


def _fake_rows(lo, hi, width):
    idx = torch.arange(lo, hi, dtype=torch.float32)[:, None]
    return idx * 10.0 + torch.arange(width, dtype=torch.float32)[None, :]


def _worker(rank, world, port, n_total, width, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi, _ = shard_bounds(n_total, rank, world)
    out = gather_rows(_fake_rows(lo, hi, width), n_total)
    dist.barrier()
    if rank == 0:
        q.put(out.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7])
def test_gather_rows_world2(n_total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got, _fake_rows(0, n_total, 5).numpy())


def test_gather_rows_single_process_is_identity():
    r = _fake_rows(0, 3, 4)
    assert gather_rows(r, 3) is r


# ---- the N > 1 logic of bench.py with a stub pipeline: real row width, NaN cells, uneven shards, both scalings ----------
ROW_WIDTH = 25 + 912 + 2 + 1        # [mshds | smile | logits | frames]  


def _stub_rows(first, n_local, pool):
    """What pipeline.run would return for the clips of one rank: a deterministic function of the pool member only,
    with NaN cells where MSHDS helpers fail (reference src/mshds_extractor.py:450-457 convention)."""
    from robust_speech_analysis_framework_amd import benchlib
    members = benchlib.pool_members(first, n_local, pool)
    m = torch.tensor(members, dtype=torch.float32)[:, None]
    rows = m * 7.0 + torch.arange(ROW_WIDTH, dtype=torch.float32)[None, :] * 0.5
    rows[:, 12] = float("nan")                                   # a column that is NaN for every clip
    rows[m[:, 0] % 3 == 0, 9] = float("nan")                     # and one that is NaN for some clips
    return rows, members


def _bench_worker(rank, world, port, scenarios, pool, q):
    from robust_speech_analysis_framework_amd import benchlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    for clips_per_gpu, total in scenarios:
        first, n_local, n_total, scaling = benchlib.shard_plan(rank, world, clips_per_gpu, total)
        rows, members = _stub_rows(first, n_local, pool)
        out = gather_rows(rows, n_total)
        local = out[first:first + n_local]
        ok = benchlib.duplicates_bit_identical(local, members) and out.shape == (n_total, ROW_WIDTH)
        t = torch.tensor([1.0 + rank], dtype=torch.float64)          # timing reduction of bench.py: MAX over ranks
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        res.append((ok, scaling, float(t.item()), out.numpy() if rank == 0 else None))
    q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,scenarios", [(2, [(5, None), (0, 7)]), (3, [(4, None), (0, 10), (0, 2)])])
def test_bench_sharding_with_stub_pipeline(world, scenarios):
    """(clips per GPU, total): weak scaling, strong scaling with an uneven last shard, and a rank that owns nothing."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pool = 3
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, scenarios, pool, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, (clips_per_gpu, total) in enumerate(scenarios):
        assert all(res[r][k][0] for r in range(world))
        assert {res[r][k][1] for r in range(world)} == {"strong" if total is not None else "weak"}
        assert {res[r][k][2] for r in range(world)} == {float(world)}   # max over ranks
        got = res[0][k][3]
        n_total = total if total is not None else clips_per_gpu * world
        want, _ = _stub_rows(0, n_total, pool)                          # global clip order, NaN cells in place
        assert got.shape == (n_total, ROW_WIDTH)
        assert np.array_equal(np.isnan(got), np.isnan(want.numpy()))
        assert np.array_equal(np.nan_to_num(got), np.nan_to_num(want.numpy()))


def test_shard_plan_c5_shape():
    from robust_speech_analysis_framework_amd import benchlib
    assert benchlib.shard_plan(7, 8, 1000, 10000) == (8750, 1250, 10000, "strong")     # BASELINE config C5
    assert benchlib.shard_plan(3, 8, 1000, None) == (3000, 1000, 8000, "weak")
    assert benchlib.shard_plan(2, 3, 0, 2) == (2, 0, 2, "strong")                       # a rank may own nothing
    rows = torch.tensor([[1.0, float("nan")], [2.0, 3.0], [1.0, float("nan")]])
    assert benchlib.duplicates_bit_identical(rows, [0, 1, 0])
    rows[2, 0] = 1.5
    assert not benchlib.duplicates_bit_identical(rows, [0, 1, 0])


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form of the driver's N = 1 command) must start two
    ranks itself - torchrun as a child process, before anything touches a GPU - and relay rank 0's JSON line; run here
    with the fabricated-rows selftest (gloo, no GPU work), strong and weak plans, and an uneven shard."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    for extra, n_total in ((["--clips", "3"], 6), (["--total-clips", "7"], 7)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                            "--launcher-selftest", *extra], capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout                                    # ONE JSON line, from rank 0
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
        assert line["config"]["clips_total"] == n_total and line["value"] is None
        assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
        g = line["checks"]["gathered"]                                      # the line proves that both ranks' rows arrived
        assert g["gathered_rows"] == n_total and g["ranks_contributing"] == 2 == g["ranks_with_a_shard"]
        assert g["rows_at_their_global_position_from_their_owner_rank"] and g["duplicate_clips_bit_identical_across_the_gathered_table"]


def test_bench_selftest_world3_uneven_shard_proves_every_rank_contributed():
    """World 3 with an uneven last shard (10 clips: 4 + 4 + 2) and with a rank that owns nothing (2 clips: 1 + 1 + 0): rank 0's
    line carries, for the GATHERED table, the owner-rank order check, the number of contributing ranks and the cross-rank
    duplicate check."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    for total, owners in ((10, 3), (2, 2)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0",
                            "--launcher-selftest", "--total-clips", str(total)], capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        g = line["checks"]["gathered"]
        assert line["n_gpus"] == 3 and g["gathered_rows"] == total
        assert g["ranks_contributing"] == owners == g["ranks_with_a_shard"]
        assert g["rows_at_their_global_position_from_their_owner_rank"] and g["duplicate_clips_bit_identical_across_the_gathered_table"]


def test_check_gathered_notices_a_missing_or_misplaced_shard():
    from robust_speech_analysis_framework_amd import benchlib
    world, per, pool = 3, 4, 3
    parts = []
    for r in range(world):
        m = torch.as_tensor(benchlib.pool_members(r * per, per, pool), dtype=torch.float32)[:, None]
        parts.append(benchlib.tag_rows(m * 2.0 + torch.arange(6, dtype=torch.float32)[None, :], r))
    good = torch.cat(parts)
    rows, g = benchlib.check_gathered(good, world, per, None, pool)
    assert rows.shape == (12, 6) and g["ranks_contributing"] == 3 and g["rows_at_their_global_position_from_their_owner_rank"]
    assert g["duplicate_clips_bit_identical_across_the_gathered_table"]
    swapped = torch.cat([parts[1], parts[0], parts[2]])                      # shards in the wrong order
    assert not benchlib.check_gathered(swapped, world, per, None, pool)[1]["rows_at_their_global_position_from_their_owner_rank"]
    lost = torch.cat([parts[0], parts[0], parts[2]])                         # rank 1's rows never arrived
    g = benchlib.check_gathered(lost, world, per, None, pool)[1]
    assert g["ranks_contributing"] == 2 and not g["rows_at_their_global_position_from_their_owner_rank"]
    parts[2][1, 0] += 1e-3                                                   # one GPU computes a different value for the same clip
    assert not benchlib.check_gathered(torch.cat(parts), world, per, None, pool)[1]["duplicate_clips_bit_identical_across_the_gathered_table"]


def test_bench_under_an_existing_launcher_does_not_spawn_again():
    """With WORLD_SIZE in the environment (torchrun's) bench.py is a rank, not a launcher."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--launcher-selftest"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "torch.distributed.run" not in r.stderr
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1
