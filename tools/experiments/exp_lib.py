"""Build and bind tools/experiments/librsaf_exp.so (EXPERIMENT kernels; never loaded by the product path)."""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
LIB = os.path.join(HERE, "librsaf_exp.so")
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float


def build():
    from robust_speech_analysis_framework_amd import build as b
    csrc = b.CSRC
    cmd = [b._hipcc(), *b.COMMON_FLAGS, "-I", HERE, "-shared", os.path.join(HERE, "gemm_bf16x6.hip"),
           os.path.join(csrc, "runtime.hip"), "-o", LIB]
    subprocess.run(cmd, check=True)
    return LIB


def load():
    if not os.path.exists(LIB):
        build()
    import torch
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB)
    lib.rsaf_split_bf16x3.argtypes = [_P, _L, _P, _P]
    lib.rsaf_gemm_f32_bf16x6.argtypes = [_P, _P, _L, _P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _I, _F, _P]
    lib.rsaf_gemm_bf16x6_presplit.argtypes = [_P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _I, _F, _P]
    return lib


if __name__ == "__main__":
    print(build())
