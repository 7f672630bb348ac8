"""Speed of rsaf_gemm_f16x3 (two fp16 planes, three products) against rsaf_gemm_f32 on the
Wav2Vec2 shapes (both operands as k16 panels, as the encoder uses them; G3_A_ROW_MAJOR=1: A row-major as the conv layers)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robust_speech_analysis_framework_amd import _lib, ops

lib = _lib.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1752 * 249
shapes = [(rows, 2304, 768, "qkv", 0, False), (rows, 768, 768, "out-proj +R", 0, True), (rows, 3072, 768, "ffn1 gelu->planes", 1, False),
          (rows, 768, 3072, "ffn2 +R", 0, True), (rows, 768, 512, "feature projection", 0, False)]
torch.manual_seed(0)


def ev(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def scales(X):
    r, k = X.shape
    s = torch.empty(r, device="cuda")
    n = torch.empty(r, device="cuda")
    _lib.check(lib.rsaf_f16x2_row_scales(_lib.ptr(X), r, k, k, _lib.ptr(s), _lib.ptr(n), None), "scales")
    return s, n


for M, N, K, tag, act, resid in shapes:
    A = torch.randn((M, K), device="cuda")
    W = torch.randn((N, K), device="cuda") / K ** 0.5
    bias = torch.randn((N,), device="cuda")
    R = torch.randn((M, N), device="cuda") if resid else None
    a_pan = os.environ.get("G3_A_ROW_MAJOR") != "1"
    sa, an = scales(A)
    sw, wn = scales(W)
    ap = torch.empty((2, M * K), dtype=torch.int16, device="cuda")
    wp = torch.empty((2, N * K), dtype=torch.int16, device="cuda")
    _lib.check(lib.rsaf_split_f16x2(_lib.ptr(W), N, K, K, _lib.ptr(sw), 1, _lib.ptr(wp), N * K, 1, None), "split")
    t_split = ev(lambda: _lib.check(lib.rsaf_split_f16x2(_lib.ptr(A), M, K, K, _lib.ptr(sa), 1, _lib.ptr(ap), M * K, int(a_pan), None), "split"))
    planes_out = act == 1
    C = None if planes_out else torch.empty((M, N), device="cuda")
    P = torch.empty((2, M * N), dtype=torch.int16, device="cuda") if planes_out else None
    cs = torch.exp2(14 - torch.floor(torch.log2(an * wn.max() + bias.abs().max())))
    t3 = ev(lambda: _lib.check(lib.rsaf_gemm_f16x3(_lib.ptr(ap), M * K, _lib.ptr(sa), 1, _lib.ptr(wp), N * K, _lib.ptr(sw),
                                                   _lib.ptr(C) if C is not None else None, _lib.ptr(P) if P is not None else None, M * N,
                                                   _lib.ptr(cs) if planes_out else None, 1, None, _lib.ptr(bias),
                                                   _lib.ptr(R) if resid else None, M, N, K, K, K, N, N, act, 1.0, int(a_pan), 1,
                                                   int(planes_out), None), "g3"))
    t32 = ev(lambda: ops.linear(A, W, bias=bias, residual=R, act="gelu" if act == 1 else None), reps=2)
    fl = 2.0 * M * N * K
    print(f"{tag:24s} M={M} N={N} K={K}: fp32 {t32:7.3f} ms {fl / t32 / 1e9:6.1f} TF | f16x3 (B panels{', A panels' if a_pan else ''}) "
          f"{t3:7.3f} ms {fl / t3 / 1e9:6.1f} TF-eq x{t32 / t3:.2f} = {3 * fl / t3 / 1e12:.3f} PFLOP/s of fp16 products "
          f"| split of A {t_split:6.3f} ms", flush=True)
    del A, W, ap, wp, C, P, R
    torch.cuda.empty_cache()
