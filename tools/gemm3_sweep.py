"""K / N sweep of rsaf_gemm_f16x3 (fp32 output, panels): time = a + b * (K / 16) per tile -> fixed cost per tile and
asymptotic rate."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robust_speech_analysis_framework_amd import _lib

lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1752 * 249


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


for N in (768, 2304):
    for K in (256, 768, 1536, 3072):
        ap = torch.randint(-3000, 3000, (2, M * K), dtype=torch.int16, device="cuda")
        wp = torch.randint(-3000, 3000, (2, N * K), dtype=torch.int16, device="cuda")
        sa = torch.ones(M, device="cuda")
        sw = torch.ones(N, device="cuda")
        C = torch.empty((M, N), device="cuda")
        t = ev(lambda: _lib.check(lib.rsaf_gemm_f16x3(_lib.ptr(ap), M * K, _lib.ptr(sa), 1, _lib.ptr(wp), N * K, _lib.ptr(sw), _lib.ptr(C),
                                                      None, 0, None, 0, None, None, None, M, N, K, K, K, N, N, 0, 1.0, 1, 1, 0, None), "g3"))
        fl = 2.0 * M * N * K
        tiles = -(-M // 256) * -(-N // 256)
        print(f"N={N} K={K}: {t:7.3f} ms {fl / t / 1e9:6.1f} TF-eq; per tile round ({tiles / 256:.1f} rounds) {1e3 * t / (tiles / 256):7.2f} us", flush=True)
        del ap, wp, C
        torch.cuda.empty_cache()
