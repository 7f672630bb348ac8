"""EXPERIMENT accuracy check (was a pytest case while the kernels lived in librsaf.so): python tools/experiments/check_bf16x6.py on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exp_lib


def check(M, N, K):
    """EXPERIMENT kernels (gemm_bf16x6.hip, not on the default path): six bf16 MFMA partial products of three-way
    splits reproduce the fp32 product to a few fp32 roundings (checked against float64, same bar as the fp32 GEMM)."""
    import torch
    from robust_speech_analysis_framework_amd import _lib, ops
    _lib.load()
    lib = exp_lib.load()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn((M, K), generator=g).cuda()
    W = (torch.randn((N, K), generator=g) / K ** 0.5).cuda()
    bias = torch.randn((N,), generator=g).cuda()
    R = torch.randn((M, N), generator=g).cuda()
    wp = torch.empty((3, N, K), dtype=torch.int16, device="cuda")
    ap = torch.empty((3, M, K), dtype=torch.int16, device="cuda")
    _lib.check(lib.rsaf_split_bf16x3(_lib.ptr(W), N * K, _lib.ptr(wp), None), "split")
    _lib.check(lib.rsaf_split_bf16x3(_lib.ptr(A), M * K, _lib.ptr(ap), None), "split")
    # the three planes add back to the float to within 2^-24
    def f32(p):
        return (p.to(torch.int32) << 16).view(torch.float32)
    back = f32(wp[0]).double() + f32(wp[1]).double() + f32(wp[2]).double()
    assert ((back - W.double()).abs() <= W.double().abs() * 2.0 ** -23).all()
    ref = torch.nn.functional.gelu(A.double() @ W.double().T + bias.double() + R.double())
    scale = ref.abs().max().item()
    out32 = ops.linear(A, W, bias=bias, residual=R, act="gelu")
    e32 = (out32.double() - ref).abs().max().item() / scale
    for variant in ("fly", "presplit"):
        out = torch.full((M, N), float("nan"), device="cuda")
        if variant == "fly":
            if K % 32:
                continue
            _lib.check(lib.rsaf_gemm_f32_bf16x6(_lib.ptr(A), _lib.ptr(wp), N * K, _lib.ptr(out), _lib.ptr(bias), _lib.ptr(R), M, N, K,
                                                K, K, N, N, 1, 1.0, None), "gemm6")
        else:
            _lib.check(lib.rsaf_gemm_bf16x6_presplit(_lib.ptr(ap), M * K, _lib.ptr(wp), N * K, _lib.ptr(out), _lib.ptr(bias), _lib.ptr(R),
                                                     M, N, K, K, K, N, N, 1, 1.0, None), "gemm6p")
        torch.cuda.synchronize()
        e = (out.double() - ref).abs().max().item() / scale
        assert e < max(3 * e32, 2e-6), (variant, e, e32)


if __name__ == "__main__":
    for shape in [(300, 200, 64), (129, 384, 96), (1000, 130, 512)]:
        check(*shape)
        print(shape, "ok")
