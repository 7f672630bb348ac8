// cContourSmoother (sma3) -> cDeltaRegression (W=2) -> cFunctionals (12 statistics over the whole
// clip, or over its first window_frames frames: the two readings of Androids.conf:349-356) for gfx950: Androids.conf:284-368 of the reference, reached through
// src/opensmile_extractor.py:62-87.
//
// One wave per (clip, LLD contour).  The contour (<= a few thousand frames, contiguous because the
// LLD buffer is contour-major) is streamed twice from L2: pass 1 = extrema with first-occurrence
// positions and the means, pass 2 = central moments and the regression sums.  Contours, smoothing, delta regression and
// all accumulation are float64, in the operation order of the oracle (sma3 = ((a + b) + c) / 3, delta = ((s1 - s-1) +
// 2 (s2 - s-2)) / 10, no fused multiply-add): contours that hold exact ties by construction (zero-crossing counts / N,
// roll-off bins * df, runs of zeros and of repeated jitter values) then tie identically on both sides, and the
// first-occurrence positions (maxPos / minPos) are bit-exact: per-lane strict comparisons in ascending frame order, then
// a (value, index) wave reduction that prefers the smaller index on ties.
//
// Semantics = oracle/smile_oracle.py (sma3 / delta2 with edge replication, population moments).
#include "rsaf_common.h"

#pragma clang fp contract(off)

namespace rsaf {
namespace smile {

constexpr int NLLD = RSAF_SMILE_NLLD;
constexpr int NFUNC = 12;

struct Ext {
    double v;
    int i;
};

__device__ __forceinline__ Ext wave_argmax(Ext a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(a.v, o, 64);
        const int oi = __shfl_xor(a.i, o, 64);
        if (ov > a.v || (ov == a.v && oi < a.i)) { a.v = ov; a.i = oi; }
    }
    return a;
}
__device__ __forceinline__ Ext wave_argmin(Ext a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(a.v, o, 64);
        const int oi = __shfl_xor(a.i, o, 64);
        if (ov < a.v || (ov == a.v && oi < a.i)) { a.v = ov; a.i = oi; }
    }
    return a;
}

__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// sma3 of contour x (length T) at index u (already clamped to [0, T-1])
__device__ __forceinline__ double sma_at(const double* __restrict__ x, int u, int T) {
    const double a = x[clampi(u - 1, T - 1)], b = x[u], c = x[clampi(u + 1, T - 1)];
    return ((a + b) + c) / 3.0;
}

__device__ __forceinline__ void sma_delta_at(const double* __restrict__ x, int t, int T, double& s, double& d) {
    const double sm2 = sma_at(x, clampi(t - 2, T - 1), T);
    const double sm1 = sma_at(x, clampi(t - 1, T - 1), T);
    s = sma_at(x, t, T);
    const double sp1 = sma_at(x, clampi(t + 1, T - 1), T);
    const double sp2 = sma_at(x, clampi(t + 2, T - 1), T);
    d = ((sp1 - sm1) + 2.0 * (sp2 - sm2)) / 10.0;
}

__device__ void write_stats(double* __restrict__ o, Ext mx, Ext mn, double mean, double m2, double m3,
                            double m4, double sty, int T) {
    const double n = (double)T;
    const double var = m2 / n;
    const double tm = 0.5 * (n - 1.0);
    const double stt = n * (n * n - 1.0) / 12.0;
    const double slope = T > 1 ? sty / stt : 0.0;
    const double icpt = mean - slope * tm;
    double errq = var - slope * slope * (stt / n);
    if (errq < 0.0) errq = 0.0;
    const double sd = sqrt(var);
    o[0] = mx.v;
    o[1] = mn.v;
    o[2] = mx.v - mn.v;
    o[3] = (double)mx.i;
    o[4] = (double)mn.i;
    o[5] = mean;
    o[6] = slope;
    o[7] = icpt;
    o[8] = errq;
    o[9] = sd;
    o[10] = var > 0.0 ? (m3 / n) / (var * sd) : 0.0;
    o[11] = var > 0.0 ? (m4 / n) / (var * var) : 0.0;
}

__global__ __launch_bounds__(256) void smile_functionals_kernel(const double* __restrict__ lld,
                                                                const int64_t* __restrict__ frame_off,
                                                                int n_clips, int64_t total_frames,
                                                                int window_frames, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= (int64_t)n_clips * NLLD) return;   // wave-uniform
    const int clip = (int)(gw / NLLD);
    const int li = (int)(gw % NLLD);
    const int64_t fo = frame_off[clip];
    const int T = (int)(frame_off[clip + 1] - fo);

    // column placement (cCsvSink order: lld, lld_de, lld2, lld2_de, lld3, lld3_de)
    int lo, nlev, base;
    if (li < 16) { lo = 0; nlev = 16; base = 0; }
    else if (li < 22) { lo = 16; nlev = 6; base = 16 * 24; }
    else { lo = 22; nlev = 16; base = 22 * 24; }
    double* o_s = out + (int64_t)clip * RSAF_SMILE_NFEAT + base + (li - lo) * NFUNC;
    double* o_d = o_s + nlev * NFUNC;

    if (T <= 0) {
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        if (lane < NFUNC) { o_s[lane] = qnan; o_d[lane] = qnan; }
        return;
    }
    const double* x = lld + (int64_t)li * total_frames + fo;
    // statistics over the first TW frames of the full-length sma / delta contours (TW = T: the whole clip)
    const int TW = window_frames > 0 ? min(window_frames, T) : T;

    // pass 1: extrema (first occurrence) and sums
    Ext smx{-INFINITY, 0x7fffffff}, smn{INFINITY, 0x7fffffff};
    Ext dmx{-INFINITY, 0x7fffffff}, dmn{INFINITY, 0x7fffffff};
    double ssum = 0.0, dsum = 0.0;
    for (int t = lane; t < TW; t += 64) {
        double s, d;
        sma_delta_at(x, t, T, s, d);
        if (s > smx.v) { smx.v = s; smx.i = t; }
        if (s < smn.v) { smn.v = s; smn.i = t; }
        if (d > dmx.v) { dmx.v = d; dmx.i = t; }
        if (d < dmn.v) { dmn.v = d; dmn.i = t; }
        ssum += s;
        dsum += d;
    }
    smx = wave_argmax(smx); smn = wave_argmin(smn);
    dmx = wave_argmax(dmx); dmn = wave_argmin(dmn);
    const double smean = wave_sum_f64(ssum) / TW;
    const double dmean = wave_sum_f64(dsum) / TW;

    // pass 2: central moments and regression cross term
    const double tm = 0.5 * ((double)TW - 1.0);
    double s2 = 0, s3 = 0, s4 = 0, sty = 0, d2 = 0, d3 = 0, d4 = 0, dty = 0;
    for (int t = lane; t < TW; t += 64) {
        double s, d;
        sma_delta_at(x, t, T, s, d);
        const double tc = (double)t - tm;
        double e = s - smean;
        double e2 = e * e;
        s2 += e2; s3 += e2 * e; s4 += e2 * e2; sty += e * tc;
        e = d - dmean;
        e2 = e * e;
        d2 += e2; d3 += e2 * e; d4 += e2 * e2; dty += e * tc;
    }
    s2 = wave_sum_f64(s2); s3 = wave_sum_f64(s3); s4 = wave_sum_f64(s4); sty = wave_sum_f64(sty);
    d2 = wave_sum_f64(d2); d3 = wave_sum_f64(d3); d4 = wave_sum_f64(d4); dty = wave_sum_f64(dty);
    if (lane == 0) {
        write_stats(o_s, smx, smn, smean, s2, s3, s4, sty, TW);
        write_stats(o_d, dmx, dmn, dmean, d2, d3, d4, dty, TW);
    }
}

}  // namespace smile
}  // namespace rsaf

using namespace rsaf;

extern "C" int rsaf_smile_functionals(const double* lld, const int64_t* frame_off, int n_clips,
                                      int64_t total_frames, int window_frames, double* out, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && window_frames >= 0, "negative n_clips or window_frames");
    if (n_clips == 0) return RSAF_OK;
    RSAF_CHECK_ARG(frame_off && out, "NULL pointer");
    RSAF_CHECK_ARG(lld || total_frames == 0, "NULL lld");
    hipStream_t s = (hipStream_t)stream;
    const int64_t waves = (int64_t)n_clips * smile::NLLD;
    const int64_t blocks = (waves + 3) / 4;
    RSAF_CHECK_ARG(blocks <= 0x7fffffffLL, "too many clips");
    ProfScope prof("smile_functionals", s, 0.0, 0.0);
    hipLaunchKernelGGL(smile::smile_functionals_kernel, dim3((unsigned)blocks), dim3(256), 0, s, lld,
                       frame_off, n_clips, total_frames, window_frames, out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}
