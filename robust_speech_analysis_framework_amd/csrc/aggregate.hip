// Session aggregation and batch assembly behind the extractors for gfx950.
//   rsaf_segment_mean_std : per participant mean and sample standard deviation (ddof = 1, NaN skipped) of every
//                           feature column, the arithmetic of DataFrame.groupby(...).agg(['mean', 'std'])
//                           in src/utils.py:49.
//   rsaf_gather_rows_f32  : row gather with zero fill: np.vstack of a participant's clip sequences
//                           (src/utils.py:96) and the right-zero-padded batch of collate_fn
//                           (src/dl_cv_strategies.py:81-84) are both one launch.
// Both are HBM-bound streaming kernels (every input element is read once or twice, coalesced along the row).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "rsaf_common.h"

namespace rsaf {
namespace aggregate {

// one workgroup per (segment, 256-column slab); thread = column, rows walked in the given order
__global__ __launch_bounds__(256) void segment_mean_std_kernel(const double* __restrict__ rows, int64_t ld,
                                                               const int* __restrict__ row_index, const int* __restrict__ seg_off,
                                                               int width, double* __restrict__ out) {
    const int seg = blockIdx.x, col = blockIdx.y * 256 + threadIdx.x;
    if (col >= width) return;
    const int a = seg_off[seg], b = seg_off[seg + 1];
    const double qn = __longlong_as_double(0x7ff8000000000000LL);
    double sum = 0.0;
    int cnt = 0;
    for (int i = a; i < b; ++i) {
        const double v = rows[(int64_t)row_index[i] * ld + col];
        if (v == v) { sum += v; ++cnt; }
    }
    const double mean = cnt > 0 ? sum / (double)cnt : qn;
    double ssq = 0.0;
    for (int i = a; i < b; ++i) {
        const double v = rows[(int64_t)row_index[i] * ld + col];
        if (v == v) { const double d = v - mean; ssq += d * d; }
    }
    double* o = out + ((int64_t)seg * width + col) * 2;
    o[0] = mean;
    o[1] = cnt > 1 ? sqrt(ssq / (double)(cnt - 1)) : qn;
}

// dst row r = src row src_row[r] (zeros when src_row[r] < 0); one wave per row, float4 when the row allows it
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t ld_src,
                                                          const int64_t* __restrict__ src_row, int64_t n_rows, int width,
                                                          float* __restrict__ dst, int64_t ld_dst, int vec4) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t s = src_row[r];
    float* d = dst + r * ld_dst;
    if (vec4) {
        const float4* sp = s >= 0 ? reinterpret_cast<const float4*>(src + s * ld_src) : nullptr;
        float4* dp = reinterpret_cast<float4*>(d);
        for (int c = lane; c < width / 4; c += 64) dp[c] = sp ? sp[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int c = lane; c < width; c += 64) d[c] = s >= 0 ? src[s * ld_src + c] : 0.0f;
    }
}

}  // namespace aggregate
}  // namespace rsaf

using namespace rsaf;

extern "C" {

int rsaf_segment_mean_std(const double* rows, int64_t ld, const int* row_index, const int* seg_off, int n_seg, int width,
                          double* out, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_seg >= 0 && width >= 0 && ld >= width, "bad sizes");
    if (n_seg == 0 || width == 0) return RSAF_OK;
    RSAF_CHECK_ARG(rows && row_index && seg_off && out, "NULL pointer");
    RSAF_CHECK_ARG(n_seg <= 0x7fffffff && (width + 255) / 256 <= 65535, "too many segments or columns for one launch");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("segment_mean_std", s, 0.0, 0.0);
    hipLaunchKernelGGL(aggregate::segment_mean_std_kernel, dim3((unsigned)n_seg, (unsigned)((width + 255) / 256)), dim3(256), 0, s,
                       rows, ld, row_index, seg_off, width, out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int rsaf_gather_rows_f32(const float* src, int64_t ld_src, const int64_t* src_row, int64_t n_rows, int width, float* dst,
                         int64_t ld_dst, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_rows >= 0 && width >= 0 && ld_src >= width && ld_dst >= width, "bad sizes");
    if (n_rows == 0 || width == 0) return RSAF_OK;
    RSAF_CHECK_ARG(src && src_row && dst, "NULL pointer");
    RSAF_CHECK_ARG((n_rows + 3) / 4 <= 0x7fffffffLL, "too many rows for one launch");
    const int vec4 = (width % 4 == 0) && (ld_src % 4 == 0) && (ld_dst % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("gather_rows", s, 0.0, 8.0 * (double)n_rows * width);
    hipLaunchKernelGGL(aggregate::gather_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, src, ld_src, src_row,
                       n_rows, width, dst, ld_dst, vec4);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

}  // extern "C"
