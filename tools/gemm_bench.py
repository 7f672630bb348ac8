"""Micro-benchmark of the fp32 MFMA GEMM at the Wav2Vec2 / CNN-LSTM shapes (HIP events)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robust_speech_analysis_framework_amd import _lib, ops

_lib.load()
shapes = [  # (M, N, K, tag)
    (4096, 4096, 4096, "square 4096"),
    (256 * 249, 2304, 768, "w2v2 qkv   (256 chunks)"),
    (256 * 249, 3072, 768, "w2v2 ffn1"),
    (256 * 249, 768, 3072, "w2v2 ffn2"),
    (256 * 249, 768, 768, "w2v2 out-proj"),
    (64 * 7999, 512, 1536, "w2v2 conv1 (64 chunks)"),
    (256 * 1500, 128, 2304, "cnn res1.conv1 (B=256)"),
    (256 * 750, 1024, 128, "lstm l0 in-proj"),
]
for M, N, K, tag in shapes:
    A = torch.randn((M, K), device="cuda")
    W = torch.randn((N, K), device="cuda")
    out = torch.empty((M, N), device="cuda")
    for _ in range(2):
        ops.linear(A, W, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        ops.linear(A, W, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{tag:28s} M={M:7d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
