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


@pytest.mark.parametrize("n_total", [7, 8, 1])
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
