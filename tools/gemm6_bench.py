"""Speed of rsaf_gemm_bf16x6 (pre-split operands) against rsaf_gemm_f32 on the Wav2Vec2 shapes, with the cost of the
split pass that a non-GEMM producer would pay (go / no-go data for replacing the fp32 GEMM in the Wav2Vec2 stage)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robust_speech_analysis_framework_amd import _lib, ops

lib = _lib.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048 * 249
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


for M, N, K, tag, act, resid in shapes:
    A = torch.randn((M, K), device="cuda")
    W = torch.randn((N, K), device="cuda") / K ** 0.5
    bias = torch.randn((N,), device="cuda")
    R = torch.randn((M, N), device="cuda") if resid else None
    ap = torch.empty((3, M, K), dtype=torch.int16, device="cuda")
    wp = torch.empty((3, N, K), dtype=torch.int16, device="cuda")
    _lib.check(lib.rsaf_split_bf16x3(_lib.ptr(W), N * K, _lib.ptr(wp), N * K, None), "split")
    t_split = ev(lambda: _lib.check(lib.rsaf_split_bf16x3(_lib.ptr(A), M * K, _lib.ptr(ap), M * K, None), "split"))
    planes_out = act == 1
    C = None if planes_out else torch.empty((M, N), device="cuda")
    P = torch.empty((3, M, N), dtype=torch.int16, device="cuda") if planes_out else None
    t6 = ev(lambda: _lib.check(lib.rsaf_gemm_bf16x6(_lib.ptr(ap), M * K, _lib.ptr(wp), N * K, _lib.ptr(C) if C is not None else None,
                                                    _lib.ptr(P) if P is not None else None, M * N, _lib.ptr(bias),
                                                    _lib.ptr(R) if resid else None, M, N, K, K, K, N, N, act, 1.0, None), "g6"))
    # the layout the Wav2Vec2 encoder uses: both operands as k16 panels
    a_pan = os.environ.get("G6_A_ROW_MAJOR") != "1"      # (G6_A_ROW_MAJOR=1: what the conv layers see: A row-major, B panels)
    wpp = torch.empty((3, N * K), dtype=torch.int16, device="cuda")
    _lib.check(lib.rsaf_split_bf16x3_panels(_lib.ptr(W), N, K, _lib.ptr(wpp), N * K, None), "split panels")
    app = ap
    if a_pan:
        app = torch.empty((3, M * K), dtype=torch.int16, device="cuda")
        _lib.check(lib.rsaf_split_bf16x3_panels(_lib.ptr(A), M, K, _lib.ptr(app), M * K, None), "split panels")
    t6p = ev(lambda: _lib.check(lib.rsaf_gemm_bf16x6_panels(_lib.ptr(app), M * K, _lib.ptr(wpp), N * K, _lib.ptr(C) if C is not None else None,
                                                            _lib.ptr(P) if P is not None else None, M * N, _lib.ptr(bias),
                                                            _lib.ptr(R) if resid else None, M, N, K, K, K, N, N, act, 1.0,
                                                            int(a_pan), 1, int(planes_out), None), "g6p"))
    del wpp, app
    t32 = ev(lambda: ops.linear(A, W, bias=bias, residual=R, act="gelu" if act == 1 else None))
    fl = 2.0 * M * N * K
    print(f"{tag:24s} M={M} N={N} K={K}: fp32 {t32:7.3f} ms {fl / t32 / 1e9:6.1f} TF | bf16x6 row-major {t6:7.3f} ms {fl / t6 / 1e9:6.1f} TF-eq "
          f"x{t32 / t6:.2f} | as used (B panels{', A panels' if a_pan else ''}) {t6p:7.3f} ms {fl / t6p / 1e9:6.1f} TF-eq x{t32 / t6p:.2f} "
          f"| split of A {t_split:6.3f} ms", flush=True)
    del A, W, ap, wp, C, P, R
    torch.cuda.empty_cache()
