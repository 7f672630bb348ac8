"""ctypes binding of librsaf.so (the C ABI in include/rsaf.h).

The product path has NO CPU fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "librsaf.so")

RSAF_OK = 0


class RsafError(RuntimeError):
    pass


class ProfRecord(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float

# symbol -> (restype, argtypes); must list every function declared in include/rsaf.h
SIGNATURES = {
    "rsaf_abi_version": (_I, []),
    "rsaf_last_error": (C.c_char_p, []),
    "rsaf_init_device": (_I, [_I]),
    "rsaf_prof_begin": (_I, []),
    "rsaf_prof_end": (_I, [C.POINTER(ProfRecord), _I, C.POINTER(_I)]),
    "rsaf_smile_geometry": (_I, [_I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "rsaf_smile_n_frames": (_L, [_L, _I]),
    "rsaf_smile_lld_batch": (_I, [_P, _P, _P, _I, _L, _L, _I, _P, _P, _P, _P]),
    "rsaf_smile_workspace_bytes": (_L, [_L]),
    "rsaf_smile_pitch_track": (_I, [_P, _P, _P, _I, _L, _I, _P, _P, _P, _P]),
    "rsaf_smile_functionals": (_I, [_P, _P, _I, _L, _I, _P, _P]),
    "rsaf_gemm_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _I, _I,
                           C.POINTER(_L), _I, _I, _F, _I, _P]),
    "rsaf_f16x2_row_scales": (_I, [_P, _L, _I, _L, _P, _P, _P]),
    "rsaf_split_f16x2": (_I, [_P, _L, _I, _L, _P, _I, _P, _L, _I, _P]),
    "rsaf_gemm_f16x3": (_I, [_P, _L, _P, _I, _P, _L, _P, _P, _P, _L, _P, _I, _P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _I, _F,
                             _I, _I, _I, _P]),
    "rsaf_cnnlstm_weight_floats": (_L, [_I, _I, _I, _I, _I]),
    "rsaf_cnnlstm_weight_offsets": (_I, [_I, _I, _I, _I, _I, C.POINTER(_L), _I, C.POINTER(_I)]),
    "rsaf_cnnlstm_workspace_bytes": (_L, [_I, _I, _I, _I, _I, _I]),
    "rsaf_cnnlstm_forward": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _L, _P, _P]),
    "rsaf_cnnlstm_forward_stages": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _L, _P, _P, _P, _P, _P, _P]),
    "rsaf_cnn_resblock_workspace_bytes": (_L, [_I, _I, _I]),
    "rsaf_cnn_resblock_forward": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P]),
    "rsaf_attnpool_forward": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "rsaf_cnnlstm_train_param_floats": (_L, [_I, _I, _I, _I, _I]),
    "rsaf_cnnlstm_train_param_offsets": (_I, [_I, _I, _I, _I, _I, C.POINTER(_L), _I, C.POINTER(_I)]),
    "rsaf_cnnlstm_train_saved_floats": (_L, [_I, _I, _I, _I, _I, _I]),
    "rsaf_cnnlstm_train_scratch_floats": (_L, [_I, _I, _I, _I, _I, _I]),
    "rsaf_cnnlstm_train_forward": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P, _P]),
    "rsaf_cnnlstm_train_backward": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P, _P]),
    "rsaf_mshds_frameout_doubles": (_I, []),
    "rsaf_mshds_clip_peak": (_I, [_P, _P, _I, _P, _P]),
    "rsaf_mshds_intensity": (_I, [_P, _P, _I, _I, _P, _I, C.c_double, _I, _P, _P, _P]),
    "rsaf_mshds_pitch_workspace_bytes_per_clip": (_L, [_I, C.POINTER(C.c_double)]),
    "rsaf_mshds_pitch": (_I, [_P, _P, _I, _I, _P, _P, _P, C.POINTER(C.c_double), _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "rsaf_mshds_pitch_dual": (_I, [_P, _P, _I, _I, _P, _P, _P, C.POINTER(C.c_double), _P, _P, _P, _P, _P, _P,
                                   C.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "rsaf_mshds_speechrate_workspace_doubles": (_L, [_I]),
    "rsaf_mshds_speechrate": (_I, [_P, _P, _I, _I, C.c_double, _P, _P, C.c_double, C.c_double, _P, _P, _P]),
    "rsaf_mshds_resample10k_table_stride": (_I, [_I]),
    "rsaf_mshds_resample10k": (_I, [_P, _P, _I, _I, _P, _I, _P, _I, _P, _P]),
    "rsaf_mshds_formants": (_I, [_P, _P, _P, _I, _I, _P, _I, C.c_double, C.c_double, C.c_double, _P, _P]),
    "rsaf_mshds_pulses_workspace_bytes": (_L, [_I, _I, C.c_double, C.c_double]),
    "rsaf_mshds_pulses": (_I, [_P, _P, _I, _I, _P, C.c_double, C.c_double, _P, _L, _P, _I, _P, _P]),
    "rsaf_mshds_ltas_slope_tilt": (_I, [_P, _P, _I, _P, _I, _P, C.c_double, C.c_double, C.c_double, _P, _P]),
    "rsaf_mshds_cpp_seg_doubles": (_I, []),
    "rsaf_mshds_cpp": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _L, _I, _P, _P]),
    "rsaf_mshds_formant_stats": (_I, [_P, _P, _I, C.c_double, _P, _I, _P, _P, _P]),
    "rsaf_mshds_hnr_mean": (_I, [_P, _P, _P, _I, _P, _P]),
    "rsaf_mshds_spectral_moments": (_I, [_P, _P, _P, _I, _I, _P, C.c_double, C.c_double, _P, _P, _I, _I, _I,
                                        C.c_double, C.c_double, _P, _P, _P]),
    "rsaf_pcm_to_mono_f32": (_I, [_P, _I, _I, _L, _P, _P]),
    "rsaf_resample_sinc_hann": (_I, [_P, _L, _P, _P, _I, _I, _I, _P, _L, _P]),
    "rsaf_praat_lowpass_batch": (_I, [_P, _P, _I, _I, C.c_double, _P, _L, _P, _P]),
    "rsaf_resample_praat_work_bytes": (_L, [_L, C.c_double, C.c_double]),
    "rsaf_praat_lowpass_max_samples": (_L, []),
    "rsaf_resample_praat": (_I, [_P, _L, C.c_double, C.c_double, _I, _P, _L, _P, _L, _P]),
    "rsaf_segment_mean_std": (_I, [_P, _L, _P, _P, _I, _I, _P, _P]),
    "rsaf_gather_rows_f32": (_I, [_P, _L, _P, _L, _I, _P, _L, _P]),
    "rsaf_w2v2_frames": (_I, [_I]),
    "rsaf_w2v2_weight_floats": (_L, [_I] * 7),
    "rsaf_w2v2_weight_offsets": (_I, [_I] * 7 + [C.POINTER(_L), _I, C.POINTER(_I)]),
    "rsaf_w2v2_workspace_bytes": (_L, [_I] * 9),
    "rsaf_w2v2_forward": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _L, _P, _P, _P]),
    "rsaf_w2v2_workspace_bytes_ragged": (_L, [C.POINTER(_I), _I] + [_I] * 7),
    "rsaf_w2v2_forward_ragged": (_I, [_P, _P, _P, C.POINTER(_I), _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _L, _P, _P, _P]),
}

_lib = None


def load():
    """Load librsaf.so (once).  Raises RsafError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RsafError(
            f"{LIB_PATH} not found: build it with "
            "`python -m robust_speech_analysis_framework_amd.build` (needs hipcc, gfx950). "
            "There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (soname
    # libamdhip64.so.7).  Load it first so librsaf.so's NEEDED libamdhip64.so.7 resolves to that
    # already-loaded copy instead of /opt/rocm's (two runtimes in one process do not share the
    # device: the second reports "no ROCm-capable device").
    import torch
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != RSAF_OK:
        msg = load().rsaf_last_error()
        raise RsafError(f"{what or 'librsaf'} failed (code {rc}): {msg.decode() if msg else ''}")


def stream_ptr(stream=None):
    """hipStream_t of a torch stream (default: torch's current stream) as void*.

    Contract of every ``stream=`` parameter of this package: the stream, if given, must BE torch's current stream (enter it
    with ``with torch.cuda.stream(s):``).  Host-built tables travel through pinned staging buffers queued on the current
    stream and temporaries return to the current stream's allocator pool, so a kernel launched on any other stream could
    read a table before its copy lands; that is refused here instead of left as a race."""
    import torch
    cur = torch.cuda.current_stream()
    if stream is not None and stream.cuda_stream != cur.cuda_stream:
        raise RsafError("stream= must be torch's current stream: enter it with `with torch.cuda.stream(s):` "
                        "(host tables are staged on the current stream)")
    return C.c_void_p(cur.cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (must be contiguous)."""
    if not t.is_contiguous():
        raise RsafError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def c_void_p_off(t, elems: int):
    """Device pointer of element ``elems`` of a contiguous torch tensor."""
    if not t.is_contiguous():
        raise RsafError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr() + int(elems) * t.element_size())


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RsafError("no HIP device visible: the MI355X path has no CPU fallback")


def prof_begin():
    check(load().rsaf_prof_begin(), "rsaf_prof_begin")


def prof_end():
    lib = load()
    cap = 64
    recs = (ProfRecord * cap)()
    n = C.c_int(0)
    check(lib.rsaf_prof_end(recs, cap, C.byref(n)), "rsaf_prof_end")
    out = {}
    for i in range(min(n.value, cap)):
        r = recs[i]
        out[r.name.decode()] = {"launches": int(r.launches), "ms": float(r.ms),
                                "flops": float(r.flops), "bytes": float(r.bytes)}
    return out
