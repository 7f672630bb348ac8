"""One-off probe (tool): what does the row-major, stride-2 A operand of the Wav2Vec2 conv layers cost against a k16-panel A?
The conv1 shape (N = 512, K = 1536, GELU -> planes) with A read (a) as the conv layers read it: rows of a [T][512] sequence at
lda = 1024 (every output row starts two input rows further), (b) as a dense row-major matrix (lda = K), (c) as k16 panels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from robust_speech_analysis_framework_amd import _lib

lib = _lib.load()
M, N, K = 2_000_000, 512, 1536
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


W = torch.randn((N, K), device="cuda") / K ** 0.5
bias = torch.randn((N,), device="cuda")
sw = torch.empty(N, device="cuda")
_lib.check(lib.rsaf_f16x2_row_scales(_lib.ptr(W), N, K, K, _lib.ptr(sw), None, None), "scales")
wp = torch.empty((2, N * K), dtype=torch.int16, device="cuda")
_lib.check(lib.rsaf_split_f16x2(_lib.ptr(W), N, K, K, _lib.ptr(sw), 1, _lib.ptr(wp), N * K, 1, None), "split")
# the sequence the strided rows come from: 2 M + 2 rows of 512 channels; planes of small integers (any fp16 pattern times alike)
plane = M * K + 4096                                  # large enough for every layout below (the strided one needs (2 M + 2) * 512)
assert plane >= (2 * M + 2) * 512
ap = torch.randint(-2000, 2000, (2, plane), dtype=torch.int16, device="cuda")
sa = torch.full((M,), 1.0, device="cuda")
cs = torch.full((M,), 2.0 ** -6, device="cuda")
P = torch.empty((2, M * N), dtype=torch.int16, device="cuda")
fl = 2.0 * M * N * K


def run(lda, a_pan):
    return ev(lambda: _lib.check(lib.rsaf_gemm_f16x3(_lib.ptr(ap), plane, _lib.ptr(sa), 1, _lib.ptr(wp), N * K, _lib.ptr(sw), None, _lib.ptr(P), M * N,
                                                     _lib.ptr(cs), 1, None, _lib.ptr(bias), None, M, N, K, lda, K, N, N, 1, 1.0, a_pan, 1, 0, None), "g3"))


for tag, lda, pan in (("strided rows, lda = 1024 (the conv layers)", 1024, 0), ("dense row-major, lda = 1536", 1536, 0), ("k16 panels", K, 1)):
    assert (M - 1) * lda + K <= plane if not pan else M * K <= plane          # host-side bounds check before the launch
    t = run(lda, pan)
    print(f"{tag:45s} {t:8.3f} ms  {fl / t / 1e9:7.1f} TFLOP/s-equivalent", flush=True)
