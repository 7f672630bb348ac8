"""Run one GEMM shape a few times (target of rocprofv3 --pmc runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robust_speech_analysis_framework_amd import _lib, ops

_lib.load()
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (63744, 3072, 768)))
A = torch.randn((M, K), device="cuda")
W = torch.randn((N, K), device="cuda")
out = torch.empty((M, N), device="cuda")
for _ in range(6):
    ops.linear(A, W, out=out)
torch.cuda.synchronize()
print("done", M, N, K)
