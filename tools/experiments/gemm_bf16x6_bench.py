"""EXPERIMENT: speed and accuracy of the bf16x6 split GEMM (gemm_bf16x6.hip, built by exp_lib.py into librsaf_exp.so) against the exact-fp32 MFMA GEMM, both
measured against a float64 reference, on the Wav2Vec2 shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exp_lib
import torch

from robust_speech_analysis_framework_amd import _lib, ops

_lib.load()
lib = exp_lib.load()
shapes = [(256 * 249, 2304, 768, "w2v2 qkv"), (256 * 249, 3072, 768, "w2v2 ffn1"), (256 * 249, 768, 3072, "w2v2 ffn2"),
          (256 * 249, 768, 768, "w2v2 out-proj"), (4096, 4096, 4096, "square 4096")]
torch.manual_seed(0)
for M, N, K, tag in shapes:
    A = torch.randn((M, K), device="cuda")
    W = torch.randn((N, K), device="cuda") / K ** 0.5
    bias = torch.randn((N,), device="cuda")
    planes = torch.empty((3, N, K), dtype=torch.int16, device="cuda")
    _lib.check(lib.rsaf_split_bf16x3(_lib.ptr(W), N * K, _lib.ptr(planes), _lib.stream_ptr(None)), "split")
    out6 = torch.empty((M, N), device="cuda")
    out32 = torch.empty((M, N), device="cuda")

    def run6():
        _lib.check(lib.rsaf_gemm_f32_bf16x6(_lib.ptr(A), _lib.ptr(planes), N * K, _lib.ptr(out6), _lib.ptr(bias), None, M, N, K,
                                            K, K, N, 0, 0, 1.0, _lib.stream_ptr(None)), "gemm6")

    def run32():
        ops.linear(A, W, bias=bias, out=out32)

    aplanes = torch.empty((3, M, K), dtype=torch.int16, device="cuda")
    out6p = torch.empty((M, N), device="cuda")

    def split_a():
        _lib.check(lib.rsaf_split_bf16x3(_lib.ptr(A), M * K, _lib.ptr(aplanes), _lib.stream_ptr(None)), "split A")

    def run6p():
        _lib.check(lib.rsaf_gemm_bf16x6_presplit(_lib.ptr(aplanes), M * K, _lib.ptr(planes), N * K, _lib.ptr(out6p), _lib.ptr(bias),
                                                 None, M, N, K, K, K, N, 0, 0, 1.0, _lib.stream_ptr(None)), "gemm6p")

    split_a()
    res = {}
    for name, fn in (("fp32 MFMA", run32), ("bf16x6", run6), ("split A", split_a), ("presplit", run6p)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / n
    # accuracy on a slab of rows against float64
    rows = slice(0, 512)
    ref = A[rows].double() @ W.double().T + bias.double()
    scale = ref.abs().max().item()
    e32 = (out32[rows].double() - ref).abs().max().item() / scale
    e6 = (out6[rows].double() - ref).abs().max().item() / scale
    e6p = (out6p[rows].double() - ref).abs().max().item() / scale
    fl = 2.0 * M * N * K
    print(f"{tag:14s} M={M:6d} N={N:5d} K={K:5d}  fp32 {res['fp32 MFMA']:7.3f} ms {fl / res['fp32 MFMA'] / 1e9:6.1f} TF/s err {e32:.2e} | "
          f"bf16x6 {res['bf16x6']:7.3f} ms {fl / res['bf16x6'] / 1e9:6.1f} TF/s-equiv err {e6:.2e} | x{res['fp32 MFMA'] / res['bf16x6']:.2f} | "
          f"presplit {res['presplit']:7.3f} ms {fl / res['presplit'] / 1e9:6.1f} TF/s-equiv (+ split A {res['split A']:.3f} ms) err {e6p:.2e} | "
          f"x{res['fp32 MFMA'] / res['presplit']:.2f} / x{res['fp32 MFMA'] / (res['presplit'] + res['split A']):.2f} incl. split", flush=True)
