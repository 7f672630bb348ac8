"""The packed GELU of the GEMM / positional-conv / conv0 epilogues (csrc/gemm_f16x3.h: gelu_pair) replayed on the host.

The device function is a straight line of fp32 multiplies and fused multiply-adds plus one exp2; this test reads its constants
out of the header, replays the same operations in numpy (every FMA as a float64 product-sum rounded once to float32 - what
v_pk_fma_f32 does) and holds the result against float64 erf: the bar is the error of the formula it replaced (x / 2 (1 +
erff(x / sqrt 2)) with a correctly rounded erff: 1.1e-7 of max(|x|, 1e-3))."""
import os
import re

import numpy as np
from scipy.special import erf

HERE = os.path.dirname(os.path.abspath(__file__))
HDR = os.path.join(HERE, "..", "robust_speech_analysis_framework_amd", "csrc", "gemm_f16x3.h")
f32 = np.float32


def _constants():
    src = open(HDR).read()
    body = src[src.index("gelu_f32x2 gelu_pair(gelu_f32x2 x) {"):]
    body = body[:body.index("#endif")]
    return [float(v) for v in re.findall(r"(-?\d+\.\d+(?:e[-+]?\d+)?)f", body)]


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def _gelu_pair_replay(x, K):
    (inv_sqrt2, c0, c1, c2, c3, c4, c5, c6, nl2e, one, d0, d1, d2, d3, d4, d5, cut_x, cut_y, half) = K
    assert one == 1.0 and cut_x == cut_y == 0.927734375 and half == 0.5
    k = lambda v: np.full(x.shape, f32(v))                                      # noqa: E731
    a = (x * f32(inv_sqrt2)).astype(f32)
    t = np.abs(a)
    s = (a * a).astype(f32)
    r = _fma(k(c0), t, k(c1))
    u = _fma(k(c2), t, k(c3))
    r = _fma(r, s, u)
    for c in (c4, c5, c6):
        r = _fma(r, t, k(c))
    r = _fma(r, t, (t * f32(nl2e)).astype(f32))
    e = np.exp2(r.astype(np.float64)).astype(f32)                               # v_exp_f32: 1 ulp
    far = np.copysign((f32(1.0) - e).astype(f32), a)
    q = _fma(k(d0), s, k(d1))
    for d in (d2, d3, d4, d5):
        q = _fma(q, s, k(d))
    near = _fma(q, a, a)
    er = np.where(t > f32(cut_x), far, near).astype(f32)
    hx = (x * f32(half)).astype(f32)
    return _fma(hx, er, hx)


def test_packed_gelu_matches_float64_as_well_as_the_library_form():
    K = _constants()
    assert len(K) == 19, K
    rng = np.random.default_rng(3)
    x = np.concatenate([np.linspace(-8.0, 8.0, 2_000_001), rng.normal(0.0, 1.5, 1_000_000), rng.normal(0.0, 0.05, 200_000)]).astype(f32)
    ref = 0.5 * x.astype(np.float64) * (1.0 + erf(x.astype(np.float64) / np.sqrt(2.0)))
    got = _gelu_pair_replay(x, K).astype(np.float64)
    err = np.abs(got - ref) / np.maximum(np.abs(x), 1e-3)
    lib = (f32(0.5) * x * (f32(1.0) + erf((x * f32(0.70710678118654752440)).astype(f32)).astype(f32))).astype(f32).astype(np.float64)
    err_lib = np.abs(lib - ref) / np.maximum(np.abs(x), 1e-3)
    assert err.max() <= 1.1e-7 and err.max() <= 1.05 * err_lib.max(), (err.max(), err_lib.max())
    edge = _gelu_pair_replay(np.array([-1e30, -40.0, -12.0, -0.0, 0.0, 12.0, 40.0, 1e30], dtype=f32), K)
    assert np.array_equal(edge, np.array([-0.0, -0.0, -0.0, -0.0, 0.0, 12.0, 40.0, 1e30], dtype=f32))
    assert np.isnan(_gelu_pair_replay(np.array([np.nan], dtype=f32), K)[0])
