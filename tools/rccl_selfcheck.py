"""Single-rank RCCL self-check of the collectives bench.py uses at N > 1 (all_gather_into_tensor of the float32 result
rows, all_reduce(MAX) of a float64 scalar, barrier): proves the calls, dtypes and shapes are accepted by the nccl (= RCCL)
backend on this image.  The multi-rank data path itself is covered by the gloo tests (tests/test_dist_gloo.py)."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rows = torch.arange(1000 * 940, dtype=torch.float32, device=dev).view(1000, 940)
out = torch.empty_like(rows)
dist.all_gather_into_tensor(out, rows)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(out, rows) and float(t) == 1.25
print("RCCL self-check ok:", dist.get_backend(), torch.cuda.get_device_name(0))
dist.destroy_process_group()
