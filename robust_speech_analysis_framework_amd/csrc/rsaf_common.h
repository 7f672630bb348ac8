// Internal helpers shared by the gfx950 kernels of librsaf.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "rsaf.h"

namespace rsaf {

void set_error(const std::string& msg);

#define RSAF_CHECK_ARG(cond, msg)                                                   \
    do {                                                                            \
        if (!(cond)) {                                                              \
            ::rsaf::set_error(std::string(__func__) + ": " + (msg));                \
            return RSAF_ERR_ARG;                                                    \
        }                                                                           \
    } while (0)

#define RSAF_CHECK_HIP(expr)                                                        \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            ::rsaf::set_error(std::string(__func__) + ": " #expr " -> " +           \
                              hipGetErrorString(_e));                               \
            return RSAF_ERR_HIP;                                                    \
        }                                                                           \
    } while (0)

// Per-kernel-family event timing (rsaf_prof_begin/end).  No-ops unless profiling is on.
struct ProfScope {
    ProfScope(const char* name, hipStream_t s, double flops, double bytes);
    ~ProfScope();
    int slot, pair;
    hipStream_t stream;
};

// ---- wave64 DPP reductions (gfx9 row_shr / row_bcast forms) -------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}

// inclusive prefix sum across the 64 lanes of a wave (lane 63 ends with the total)
__device__ __forceinline__ float wave_scan_incl(float x) {
    x += dpp_f32<0x111>(x);         // row_shr:1
    x += dpp_f32<0x112>(x);         // row_shr:2
    x += dpp_f32<0x114>(x);         // row_shr:4
    x += dpp_f32<0x118>(x);         // row_shr:8
    x += dpp_f32<0x142, 0xA>(x);    // row_bcast:15 -> rows 1,3
    x += dpp_f32<0x143, 0xC>(x);    // row_bcast:31 -> rows 2,3
    return x;
}

__device__ __forceinline__ float wave_sum(float x) {
    x = wave_scan_incl(x);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__device__ __forceinline__ int wave_min_i32(int x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x = min(x, __shfl_xor(x, o, 64));
    return x;
}

__device__ __forceinline__ float readlane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// wave maximum of non-negative values (the zero fill of the DPP shifts is neutral), broadcast to every lane
__device__ __forceinline__ float wave_max_nonneg(float x) {
    x = fmaxf(x, dpp_f32<0x111>(x));
    x = fmaxf(x, dpp_f32<0x112>(x));
    x = fmaxf(x, dpp_f32<0x114>(x));
    x = fmaxf(x, dpp_f32<0x118>(x));
    x = fmaxf(x, dpp_f32<0x142, 0xA>(x));
    x = fmaxf(x, dpp_f32<0x143, 0xC>(x));
    return readlane_f32(x, 63);
}

}  // namespace rsaf
