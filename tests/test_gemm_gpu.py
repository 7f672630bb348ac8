"""fp32 MFMA GEMM vs a float64 numpy reference (through the C ABI)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(got, ref):
    return np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 96), (1, 32, 64), (257, 48, 36), (513, 768, 512)])
def test_plain_nt_asymmetric(rsaf_lib, M, N, K):
    """C = A @ W^T with asymmetric operands (catches a transposed C/D map)."""
    import torch
    from robust_speech_analysis_framework_amd import ops
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)
    W[:, 0] += np.arange(N, dtype=np.float32)          # strongly asymmetric
    out = ops.linear(torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda())
    torch.cuda.synchronize()
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    assert _rel(out.cpu().numpy(), ref) < 5e-6


def test_identity_a_returns_b_transposed(rsaf_lib):
    import torch
    from robust_speech_analysis_framework_amd import ops
    n = 96
    A = np.eye(n, dtype=np.float32)
    W = np.arange(n * n, dtype=np.float32).reshape(n, n)   # exact integers, asymmetric
    out = ops.linear(torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), W.T)


@pytest.mark.parametrize("act", [None, "gelu", "silu"])
def test_bias_residual_activation(rsaf_lib, act):
    import torch
    from robust_speech_analysis_framework_amd import ops
    rng = np.random.default_rng(5)
    M, N, K = 200, 130, 64          # N not a multiple of 4: plain [N,K] weights allow it
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 8).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    out = ops.linear(torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda(), torch.from_numpy(b).cuda(),
                     act=act, residual=torch.from_numpy(R).cuda())
    torch.cuda.synchronize()
    ref = A.astype(np.float64) @ W.astype(np.float64).T + b + R
    if act == "gelu":
        ref = _gelu(ref)
    elif act == "silu":
        ref = ref / (1.0 + np.exp(-ref))
    assert _rel(out.cpu().numpy(), ref) < 2e-6


def test_two_level_batch_attention_shapes(rsaf_lib):
    """S = scale * Q K^T and O = P V with (chunk, head) batches over a packed qkv buffer."""
    import torch
    from robust_speech_analysis_framework_amd import ops
    rng = np.random.default_rng(11)
    nc, nh, T, hd = 3, 4, 249, 64
    D = nh * hd
    qkv = rng.standard_normal((nc, T, 3 * D)).astype(np.float32)
    dq = torch.from_numpy(qkv).cuda()
    Tp = 256
    S = torch.zeros((nc, nh, T, Tp), dtype=torch.float32, device="cuda")
    ops.gemm_f32(dq, dq, S, T, T, hd, 3 * D, 3 * D, Tp, nz=nc * nh, nz2=nh,
                 strides=[T * 3 * D, hd, T * 3 * D, hd, nh * T * Tp, T * Tp, 0, 0],
                 alpha=0.125, a_off=0, b_off=D)
    torch.cuda.synchronize()
    q = qkv[:, :, :D].reshape(nc, T, nh, hd).transpose(0, 2, 1, 3).astype(np.float64)
    k = qkv[:, :, D:2 * D].reshape(nc, T, nh, hd).transpose(0, 2, 1, 3).astype(np.float64)
    v = qkv[:, :, 2 * D:].reshape(nc, T, nh, hd).transpose(0, 2, 1, 3).astype(np.float64)
    ref = 0.125 * q @ k.transpose(0, 1, 3, 2)
    got = S.cpu().numpy()
    assert _rel(got[..., :T], ref) < 2e-6
    assert (got[..., T:] == 0).all()                     # pad columns untouched
    # O = P V, B operand [K,N] (N contiguous), K = 249 is not a multiple of 4
    P = torch.softmax(S[..., :T], dim=-1)
    Pp = torch.zeros_like(S)
    Pp[..., :T] = P
    O = torch.zeros((nc, T, D), dtype=torch.float32, device="cuda")
    ops.gemm_f32(Pp, dq, O, T, hd, T, Tp, 3 * D, D, nz=nc * nh, nz2=nh,
                 strides=[nh * T * Tp, T * Tp, T * 3 * D, hd, T * D, hd, 0, 0], b_kn=True, b_off=2 * D)
    torch.cuda.synchronize()
    refO = (P.cpu().numpy().astype(np.float64) @ v).transpose(0, 2, 1, 3).reshape(nc, T, D)
    assert _rel(O.cpu().numpy(), refO) < 2e-6


@pytest.mark.parametrize("T", [1, 2, 37, 300])
def test_conv3_pad1_in_place(rsaf_lib, T):
    """k=3/pad=1 Conv1d over a channels-last batch, input read in place (a_pad_k)."""
    import torch
    import torch.nn.functional as F
    from robust_speech_analysis_framework_amd import ops
    rng = np.random.default_rng(T)
    Bn, Cin, Cout = 3, 32, 48
    x = rng.standard_normal((Bn, T, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3)) / 4).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dx = torch.from_numpy(x).cuda()
    wk = torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 1)).reshape(Cout, 3 * Cin)).cuda()
    out = torch.empty((Bn, T, Cout), dtype=torch.float32, device="cuda")
    ops.gemm_f32(dx, wk, out, T, Cout, 3 * Cin, Cin, 3 * Cin, Cout, bias=torch.from_numpy(b).cuda(),
                 nz=Bn, nz2=1, strides=[T * Cin, 0, 0, 0, T * Cout, 0, 0, 0], a_pad_k=Cin, a_off=-Cin)
    torch.cuda.synchronize()
    ref = F.conv1d(torch.from_numpy(x).double().permute(0, 2, 1), torch.from_numpy(w).double(),
                   torch.from_numpy(b).double(), padding=1).permute(0, 2, 1).numpy()
    assert _rel(out.cpu().numpy(), ref) < 2e-6


def test_strided_conv_as_gemm(rsaf_lib):
    """k=3/stride=2 Conv1d (Wav2Vec2 feature encoder) = GEMM with lda = 2*Cin, K = 3*Cin."""
    import torch
    import torch.nn.functional as F
    from robust_speech_analysis_framework_amd import ops
    rng = np.random.default_rng(3)
    nb, Tin, Cc = 2, 401, 64
    Tout = (Tin - 3) // 2 + 1
    x = rng.standard_normal((nb, Tin, Cc)).astype(np.float32)
    w = (rng.standard_normal((Cc, Cc, 3)) / 8).astype(np.float32)
    dx = torch.from_numpy(x).cuda()
    wk = torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 1)).reshape(Cc, 3 * Cc)).cuda()
    out = torch.empty((nb, Tout, Cc), dtype=torch.float32, device="cuda")
    ops.gemm_f32(dx, wk, out, Tout, Cc, 3 * Cc, 2 * Cc, 3 * Cc, Cc, nz=nb, nz2=1,
                 strides=[Tin * Cc, 0, 0, 0, Tout * Cc, 0, 0, 0], act="gelu")
    torch.cuda.synchronize()
    ref = F.conv1d(torch.from_numpy(x).double().permute(0, 2, 1), torch.from_numpy(w).double(), stride=2)
    ref = F.gelu(ref).permute(0, 2, 1).numpy()
    assert _rel(out.cpu().numpy(), ref) < 2e-6


def test_argument_validation(rsaf_lib):
    import torch
    from robust_speech_analysis_framework_amd import _lib, ops
    a = torch.zeros((8, 6), device="cuda")
    w = torch.zeros((4, 6), device="cuda")
    with pytest.raises(_lib.RsafError):
        ops.linear(a, w)                                   # K = 6 is not a multiple of 4


# ---- fp32-accurate GEMM on the fp16 matrix pipe (three products of two-way fp16 splits, power-of-two row scales) --------
def _h3_scales(X, loose=1.0):
    """Per-row power-of-two scales from rsaf_f16x2_row_scales (+ row norms); loose > 1 shrinks them as a loose bound would."""
    import torch
    from robust_speech_analysis_framework_amd import _lib
    lib = _lib.load()
    rows, k = X.shape
    s = torch.empty(rows, device="cuda")
    nrm = torch.empty(rows, device="cuda")
    _lib.check(lib.rsaf_f16x2_row_scales(_lib.ptr(X), rows, k, k, _lib.ptr(s), _lib.ptr(nrm), None), "row scales")
    return s / loose, nrm


def _h3_split(X, s, panels):
    import torch
    from robust_speech_analysis_framework_amd import _lib
    lib = _lib.load()
    rows, k = X.shape
    P = torch.zeros((2, rows * k), dtype=torch.int16, device="cuda")
    _lib.check(lib.rsaf_split_f16x2(_lib.ptr(X), rows, k, k, _lib.ptr(s), 1, _lib.ptr(P), rows * k, int(panels), None), "split f16x2")
    return P


def _h3_to_f64(P, s, rows, k, panels):
    """planes -> the float64 values they stand for (hi + lo) / scale, as [rows][k]"""
    import torch
    v = P.view(torch.float16).double()
    v = v[0] + v[1]
    v = v.view(k // 16, rows, 16).permute(1, 0, 2).reshape(rows, k) if panels else v.view(rows, k)
    return v / s.double()[:, None]


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(300, 208, 64), (256, 256, 16), (1000, 144, 512), (513, 768, 3072), (257, 2304, 768), (700, 48, 6144)])
@pytest.mark.parametrize("loose", [1.0, 4096.0])
def test_f16x3_gemm_is_fp32_accurate(rsaf_lib, M, N, K, loose):
    """Every output mode of rsaf_gemm_f16x3 against float64: the error is that of an fp32 FMA chain (same bar as, and
    compared with, rsaf_gemm_f32 on the same operands), with exact row scales and with scales 4 096 x smaller (what a loose
    Cauchy-Schwarz bound on a GEMM output gives: the low plane is then largely subnormal); edges (M, N not multiples of
    the tile), both operand layouts and the plane output (scaled by its own c_scale) included."""
    import torch
    from robust_speech_analysis_framework_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn((M, K), generator=g)
    A[:, :3] *= 40.0                                                    # outlier channels, as transformer activations have
    A = A.cuda()
    W = (torch.randn((N, K), generator=g) / K ** 0.5).cuda()
    bias = torch.randn((N,), generator=g).cuda()
    R = torch.randn((M, N), generator=g).cuda()
    sa, anorm = _h3_scales(A, loose)
    sw, wnorm = _h3_scales(W)
    lin = A.double() @ W.double().T + bias.double()
    ref_plain, ref_res, ref_gelu = lin, lin + R.double(), torch.nn.functional.gelu(lin)
    ref_silu = lin * torch.sigmoid(lin)
    e32 = (ops.linear(A, W, bias=bias).double() - ref_plain).abs().max().item() / ref_plain.abs().max().item()
    bar = max(3 * e32, 2e-6)
    # the scale of a plane OUTPUT comes from a bound on the row: |x_mn| <= |a_m| |w_n| + |b_n|
    bound = anorm * wnorm.max() + bias.abs().max()
    cs = torch.exp2(14 - torch.floor(torch.log2(bound)))
    results = {}
    for panels in (False, True):
        ap, wp = _h3_split(A, sa, panels), _h3_split(W, sw, panels)
        torch.cuda.synchronize()
        back = _h3_to_f64(wp, sw, N, K, panels)
        assert ((back - W.double()).abs() <= W.double().abs() * 2.0 ** -21 + W.abs().max().item() * 2.0 ** -38).all()

        def run(act, want_f32, want_planes, resid, c_panels=False):
            C = torch.full((M, N), float("nan"), device="cuda") if want_f32 else None
            P = torch.zeros((2, M * N), dtype=torch.int16, device="cuda") if want_planes else None
            amax = torch.zeros(1, dtype=torch.int32, device="cuda")
            _lib.check(lib.rsaf_gemm_f16x3(_lib.ptr(ap), M * K, _lib.ptr(sa), 1, _lib.ptr(wp), N * K, _lib.ptr(sw),
                                           _lib.ptr(C) if want_f32 else None, _lib.ptr(P) if want_planes else None, M * N,
                                           _lib.ptr(cs) if want_planes else None, 1, _lib.ptr(amax), _lib.ptr(bias),
                                           _lib.ptr(R) if resid else None, M, N, K, K, K, N, N, act, 1.0, int(panels), int(panels),
                                           int(c_panels), None), "gemm3")
            torch.cuda.synchronize()
            return C, P, amax.view(torch.float32).item()

        C, _, amax = run(0, True, False, False)
        assert (C.double() - ref_plain).abs().max().item() / ref_plain.abs().max().item() < bar
        assert amax == C.abs().max().item()                                 # the reported maximum is the maximum written
        results[panels] = C
        C, _, _ = run(0, True, False, True)
        assert (C.double() - ref_res).abs().max().item() / ref_res.abs().max().item() < bar
        C, _, _ = run(1, True, False, False)
        assert (C.double() - ref_gelu).abs().max().item() / ref_gelu.abs().max().item() < bar
        C, _, _ = run(2, True, False, False)
        assert (C.double() - ref_silu).abs().max().item() / ref_silu.abs().max().item() < bar
        _, P, amax = run(1, False, True, False, c_panels=panels)
        got = _h3_to_f64(P, cs, M, N, panels)
        assert torch.isfinite(got).all()                                    # the bound held: no fp16 overflow
        assert (got - ref_gelu).abs().max().item() / ref_gelu.abs().max().item() < bar
        assert abs(amax - got.abs().max().item()) <= 1e-6 * amax
        C, P, _ = run(0, True, True, False, c_panels=panels)
        assert (_h3_to_f64(P, cs, M, N, panels) - C.double()).abs().max().item() <= 2.0 ** -21 * C.abs().max().item()
    assert torch.equal(results[False], results[True])                   # the panel layout is a permutation: same bits
    with pytest.raises(_lib.RsafError):
        lib_rc = lib.rsaf_gemm_f16x3(_lib.ptr(ap), M * K, _lib.ptr(sa), 1, _lib.ptr(wp), N * K, _lib.ptr(sw), None, None, 0, None, 0,
                                     None, None, None, M, N, K, K, K, N, N, 0, 1.0, 1, 1, 0, None)
        _lib.check(lib_rc, "no output")
