"""Thin host wrappers over the C-ABI compute primitives (device tensors in, device tensors out)."""
from __future__ import annotations

import ctypes as C

from . import _lib

ACT = {None: 0, "none": 0, "gelu": 1, "silu": 2}


def gemm_f32(A, B, C_out, M, N, K, lda, ldb, ldc, *, bias=None, residual=None, ldr=0,
             nz=1, nz2=1, strides=None, a_pad_k=0, act=None, alpha=1.0, b_kn=False,
             a_off=0, b_off=0, c_off=0, r_off=0, stream=None):
    """Raw strided/batched GEMM (see include/rsaf.h: rsaf_gemm_f32).  Offsets are in elements."""
    lib = _lib.load()
    st = None
    if strides is not None:
        st = (C.c_int64 * 8)(*[int(s) for s in strides])
    rc = lib.rsaf_gemm_f32(
        _lib.c_void_p_off(A, a_off), _lib.c_void_p_off(B, b_off), _lib.c_void_p_off(C_out, c_off),
        _lib.ptr(bias) if bias is not None else None,
        _lib.c_void_p_off(residual, r_off) if residual is not None else None,
        int(M), int(N), int(K), int(lda), int(ldb), int(ldc), int(ldr), int(nz), int(nz2), st,
        int(a_pad_k), ACT[act], float(alpha), 1 if b_kn else 0, _lib.stream_ptr(stream))
    _lib.check(rc, "rsaf_gemm_f32")
    return C_out


def linear(x, weight, bias=None, *, act=None, residual=None, out=None, stream=None):
    """y = act(x @ weight.T + bias + residual); x [M,K] contiguous, weight [N,K] contiguous."""
    import torch
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    return gemm_f32(x, weight, out, M, N, K, x.stride(0), weight.stride(0), out.stride(0), bias=bias,
                    residual=residual, ldr=(residual.stride(0) if residual is not None else 0),
                    act=act, stream=stream)
