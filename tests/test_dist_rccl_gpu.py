"""The terminal all-gather on RCCL itself: a one-GPU box cannot form a second rank, so the collective is forced on a
one-rank `nccl` group (RSAF_FORCE_COLLECTIVE=1), on the real result-row buffers (940 float32 columns with NaN cells)."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def test_forced_all_gather_runs_on_rccl_with_one_rank(rsaf_lib, monkeypatch):
    import torch
    import torch.distributed as dist
    from robust_speech_analysis_framework_amd.dist import gather_rows
    assert torch.cuda.is_available()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rows = torch.randn(37, 940, device=dev)
        rows[3, 5] = float("nan")
        monkeypatch.setenv("RSAF_FORCE_COLLECTIVE", "0")
        assert gather_rows(rows, 37) is rows                                   # one rank: nothing to exchange
        monkeypatch.setenv("RSAF_FORCE_COLLECTIVE", "1")
        calls = []
        real = dist.all_gather_into_tensor
        monkeypatch.setattr(dist, "all_gather_into_tensor", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        out = gather_rows(rows, 37)
        torch.cuda.synchronize()
        assert calls == [1] and out is not rows and out.data_ptr() != rows.data_ptr()
        assert dist.get_backend() == "nccl"
        assert torch.equal(torch.nan_to_num(out), torch.nan_to_num(rows)) and torch.isnan(out[3, 5])
    finally:
        dist.destroy_process_group()
