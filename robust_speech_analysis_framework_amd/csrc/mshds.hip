// Praat-style analyses behind the MSHDS features, float64 kernels for gfx950.
//
// Replaces the parselmouth/Praat calls of src/mshds_extractor.py (all ten helpers; the cepstral part lives in
// mshds_cpp.hip):
//   _speechrate (:11-125), _pitch_values (:127-162), _extract_pitch (:164-183), _extract_intensity (:185-205),
//   _extract_harmonicity (:207-225), _extract_Slope_Tilt (:227-251), _measureFormants (:303-338),
//   _extract_Spectral_Moments (:340-376).
// Algorithms: Boersma (1993) autocorrelation / cross-correlation pitch with sinc-interpolated candidates and the
// Viterbi path finder; Praat's intensity (Kaiser-weighted mean square), Gaussian-window spectrogram + spectral
// moments, cc pulse walker, pitch-corrected Ltas, Burg formants, de Jong & Wempe syllable nuclei.  Semantics =
// oracle/mshds_oracle.py (parity unpinned: Praat itself is not available).  Praat computes in double, so do these
// kernels (MI355X: 78 TFLOP/s fp64, vector and matrix alike); discrete decisions (voicing, path) then agree with
// the oracle.
//
// Mapping: one 256-thread workgroup per analysis frame for the pitch kernel (frame staged in LDS, correlation on
// v_mfma_f64_16x16x4_f64, candidates refined by Brent's method on a Chebyshev form of the sinc interpolation or,
// where the depth is clipped, on wave-cooperative sinc sums), one wave per frame for intensity, one wave per clip
// for the path finder and the per-clip statistics, one wave per voiced stretch for the pulse walker.
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "praat_interp.h"
#include "rsaf_common.h"
#include "wave_fft.h"

// Frame times sit exactly on half-sample positions, where Praat's nearest/low index rounding is
// decided by the last bit: evaluate t1 + f*dt etc. as separately rounded IEEE operations (no FMA
// contraction), exactly like the float64 host arithmetic of the oracle.
#pragma clang fp contract(off)

namespace rsaf {
namespace mshds {

constexpr double DXS = 1.0 / 16000.0;
constexpr double PI = 3.14159265358979323846;
constexpr int MAXC = 16;            // candidate slots per frame (max_candidates <= 15)
constexpr int MAX_MAXIMA = 96;      // local maxima considered per frame (in ascending lag order)
constexpr double GOLD = 0.38196601125010515180;   // (3 - sqrt 5) / 2

struct ClipInfo {       // one entry per clip of a launch (host-built)
    int64_t sample_off;
    int64_t frame_off;  // first frame of this clip in the per-launch frame buffers
    double t1;          // time of the first frame
    int n_samples;
    int n_frames;
    double x1;          // time of the first sample (0.5 dx for a sound read from a 16 kHz file; Praat's centred grid after Sound_resample)
    double xmax;        // end of the sound's time domain [0, xmax] (n dx for a file; the ORIGINAL duration after Sound_resample)
};

struct PitchParams {
    double dt, min_pitch, ceiling, voicing_thr, octave_cost, dt_window;
    double refine_margin;   // > 0: only candidates within this margin of the best first-pass strength are refined
    int nsamp_window, half_window, nsamp_period, half_period, min_lag, max_lag, brent_ixmax, max_cand;
    int refine_depth, is_cc;
    int nfft;               // AC: FFT length, the smallest power of two >= 1.5 nsamp_window (Praat's nsampFFT)
    double voicing_thr2;    // >= 0: also emit the candidate lists for this (lower) voicing threshold into out2
    int debug_stop;         // profiling aid (env RSAF_PITCH_STOP): leave the frame kernel after phase k; 0 = run all
    int cheb_all_full;      // no candidate of this analysis can have its interpolation depth clipped by the array ends
    int cheb_clipped;       // the Chebyshev table is followed by the tables of the clipped depths 1 .. refine_depth - 1
};

// Sampled_xToLowIndex / xToNearestIndex / xToHighIndex of the sound (0-based), x1 = time of its first sample
// (Praat rounds the 1-based real index (x - x1) / dx + 1; the + 1.0 stays a separately rounded operation: fp contract is off)
__device__ __forceinline__ int64_t low_index(double t, double x1) { return (int64_t)floor((t - x1) / DXS + 1.0) - 1; }
__device__ __forceinline__ int64_t nearest_index(double t, double x1) { return (int64_t)floor(((t - x1) / DXS + 1.0) + 0.5) - 1; }
__device__ __forceinline__ int64_t high_index(double t, double x1) { return (int64_t)ceil((t - x1) / DXS + 1.0) - 1; }

__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- per-clip mean and global peak |x - mean| ------------------------------------------------------
__global__ __launch_bounds__(256) void clip_peak_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                        double* __restrict__ gpeak) {
    __shared__ double red[4];
    __shared__ double bc;
    const ClipInfo c = ci[blockIdx.x];
    const float* x = wav + c.sample_off;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // (these loops keep eight loads in flight: an exec-masked loop with one load per turn is not unrolled by the compiler
    // and waits for the memory once per turn; the additions keep their order)
    double s = 0.0;
    const int last = c.n_samples - 1;
    for (int i0 = threadIdx.x; i0 < c.n_samples; i0 += 8 * 256) {
        float q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = x[i0 + 256 * u <= last ? i0 + 256 * u : last];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + 256 * u <= last) s += (double)q[u];
    }
    s = wave_sum_f64(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) bc = c.n_samples > 0 ? (red[0] + red[1] + red[2] + red[3]) / c.n_samples : 0.0;
    __syncthreads();
    const double mean = bc;
    double m = 0.0;
    for (int i0 = threadIdx.x; i0 < c.n_samples; i0 += 8 * 256) {
        float q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = x[i0 + 256 * u <= last ? i0 + 256 * u : last];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + 256 * u <= last) m = fmax(m, fabs((double)q[u] - mean));
    }
    m = wave_max_f64(m);
    __syncthreads();
    if (lane == 0) red[w] = m;
    __syncthreads();
    if (threadIdx.x == 0) gpeak[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// ---- intensity: one wave per frame -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void intensity_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                        const double* __restrict__ win, int half, double dt,
                                                        int subtract_mean, double* __restrict__ out) {
    const ClipInfo c = ci[blockIdx.y];
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= c.n_frames) return;
    const int lane = threadIdx.x & 63;
    const float* x = wav + c.sample_off;
    const double t = c.t1 + f * dt;
    const int64_t mid = nearest_index(t, c.x1);
    const int64_t lo = mid - half < 0 ? 0 : mid - half;
    const int64_t hi = mid + half > c.n_samples - 1 ? c.n_samples - 1 : mid + half;
    double mean = 0.0;
    if (subtract_mean) {
        double s = 0.0;
        for (int64_t i0 = lo + lane; i0 <= hi; i0 += 8 * 64) {           // eight loads in flight (see clip_peak_kernel)
            float q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = x[i0 + 64 * u <= hi ? i0 + 64 * u : hi];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + 64 * u <= hi) s += (double)q[u];
        }
        mean = wave_sum_f64(s) / (double)(hi - lo + 1);
    }
    double sw = 0.0, sx = 0.0;
    for (int64_t i0 = lo + lane; i0 <= hi; i0 += 8 * 64) {
        float q[8];
        double wq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = i0 + 64 * u <= hi ? i0 + 64 * u : hi;
            q[u] = x[i];
            wq[u] = win[i - mid + half];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + 64 * u <= hi) {
                const double d = (double)q[u] - mean;
                sw += wq[u];
                sx += d * d * wq[u];
            }
        }
    }
    sw = wave_sum_f64(sw);
    sx = wave_sum_f64(sx);
    if (lane == 0) {
        const double v = sx / sw / 4.0e-10;
        out[c.frame_off + f] = v < 1e-30 ? -300.0 : 10.0 * log10(v);
    }
}

// cos(x) for x in [0, pi] (all arguments of the sinc window are): fold to [0, pi/2] and evaluate the
// degree-18 Taylor polynomial in x^2 (remainder (pi/2)^20/20! = 3.4e-15).  The library cos/sincos cost
// ~1k cycles each in fp64 and dominated this kernel; this is ~12 FMAs.
__device__ __forceinline__ double cos_0_pi(double x) {
    const bool hi = x > 0.5 * PI;
    const double y = hi ? PI - x : x;
    const double z = y * y;
    double p = -1.0 / 6402373705728000.0;              // -1/18!
    p = p * z + 1.0 / 20922789888000.0;                // 1/16!
    p = p * z - 1.0 / 87178291200.0;                   // -1/14!
    p = p * z + 1.0 / 479001600.0;                     // 1/12!
    p = p * z - 1.0 / 3628800.0;                       // -1/10!
    p = p * z + 1.0 / 40320.0;                         // 1/8!
    p = p * z - 1.0 / 720.0;                           // -1/6!
    p = p * z + 1.0 / 24.0;
    p = p * z - 0.5;
    p = p * z + 1.0;
    return hi ? -p : p;
}
__device__ __forceinline__ double sin_0_pi(double x) { return cos_0_pi(fabs(0.5 * PI - x)); }

// 1/d for d > 0: hardware reciprocal estimate + two Newton steps (full double accuracy, ~5 ops instead
// of the ~15-op IEEE division sequence)
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = r * (2.0 - d * r);
    r = r * (2.0 - d * r);
    return r;
}

__device__ __forceinline__ double readlane_f64(double v, int l) {      // l must be wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Sum over each aligned group of G lanes, result in every lane of the group.  The 16-lane part is four DPP
// steps (xor 1, xor 2, half-row mirror, row mirror: VALU latency, no LDS crossbar); rows are then combined
// through scalar registers.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);     // row_half_mirror
    v += dpp_f64<0x140>(v);     // row_mirror
    if (G == 16) return v;
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    if (G == 64) return (r0 + r1) + (r2 + r3);
    return (threadIdx.x & 32) ? r2 + r3 : r0 + r1;
}

// maximum over the wave in every lane, same DPP / readlane structure (a ds_bpermute butterfly costs six LDS-crossbar
// round trips per reduction: in the pitch frame kernels, three reductions per frame, that was a third of the time in
// front of the correlation)
__device__ __forceinline__ double wave_max_dpp(double v) {
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return fmax(fmax(r0, r1), fmax(r2, r3));
}

// ---- sinc interpolation of an LDS array by a G-lane group (Praat NUM_interpolate_sinc) -----------------
// y: n samples (0-based); x: 0-based real position; only indices in [nz_lo, nz_hi] can be non-zero.
// Every lane of the wave must call this (the 64/G groups of a wave evaluate different x).
// RECUR: the raised-cosine window angle advances by a fixed step per term, so each lane rotates
// (cos, sin) by the group stride instead of evaluating the polynomial per term (pays for long kernels).
template <int G, bool RECUR>
__device__ double sinc_group(const double* __restrict__ y, int n, double x, int depth, int nz_lo, int nz_hi, int lg) {
    const double x1 = x + 1.0;
    const int midleft = (int)floor(x1), midright = midleft + 1;
    const bool special = (x1 > n) | (x1 < 1) | (x1 == (double)midleft);
    int si = x1 > n ? n - 1 : (x1 < 1 ? 0 : midleft - 1);
    si = si < 0 ? 0 : (si > n - 1 ? n - 1 : si);
    int d = depth;
    if (d > midright - 1) d = midright - 1;
    if (d > n - midleft) d = n - midleft;
    if (d < 0 || special) d = 0;
    const int left = midright - d, right = midleft + d;
    double acc = 0.0;
    const double a0l = PI * (x1 - midleft);               // in (0, pi) unless special
    const double hs = special ? 0.0 : 0.5 * sin_0_pi(a0l); // sin(pi - a) = sin(a): same for both halves
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // left half: 1-based ix = midleft - k; right half: ix = midright + k; k = 0..d-1.  The window
        // angle (a0 + pi k) / den stays in (0, pi).  [kmin, kmax) drops the all-zero parts of y.
        const double a0 = half == 0 ? a0l : PI * (midright - x1);
        const double iden = fast_rcp(half == 0 ? x1 - left + 1.0 : right - x1 + 1.0);
        int kmin, kmax = d;
        if (half == 0) {
            kmin = midleft - 1 - nz_hi;
            if (midleft - kmax < nz_lo) kmax = midleft - nz_lo;
        } else {
            kmin = nz_lo - midright + 1;
            if (midright + kmax - 2 > nz_hi) kmax = nz_hi - midright + 2;
        }
        kmin = kmin < 0 ? 0 : kmin;
        const int k0 = kmin + lg;
        // G is even: every term of a lane has the sign of its first one, (-1)^k0 hs is applied once behind the loop
        double part = 0.0;
        if (RECUR) {
            const double th = (a0 + PI * k0) * iden, st = (PI * G) * iden;   // st < pi whenever a lane has 2+ terms
            double c = cos_0_pi(fmin(th, PI)), sn = sin_0_pi(fmin(th, PI));
            const double C = cos_0_pi(fmin(st, PI)), S = sin_0_pi(fmin(st, PI));
            for (int k = k0; k < kmax; k += G) {
                const int idx = half == 0 ? midleft - k - 1 : midright + k - 1;
                const double a = a0 + PI * k;
                part += y[idx] * (fast_rcp(a) * (1.0 + c));
                const double c2 = c * C - sn * S;
                sn = sn * C + c * S;
                c = c2;
            }
        } else {
            for (int k = k0; k < kmax; k += G) {
                const int idx = half == 0 ? midleft - k - 1 : midright + k - 1;
                const double a = a0 + PI * k;
                part += y[idx] * (fast_rcp(a) * (1.0 + cos_0_pi(a * iden)));
            }
        }
        acc += ((k0 & 1) ? -hs : hs) * part;
    }
    acc = group_sum<G>(acc);
    return special ? y[si] : acc;
}

// Praat NUMimproveMaximum (sinc): Brent's minimiser in the netlib fminbr form on -sinc over [ix-1, ix+1],
// tolerance sqrt(eps)*|x| + tol/3 on the 1-based position, <= 60 iterations.  One G-lane group per
// candidate; `live` = this group holds a real candidate (others just keep the wave's shuffles uniform).
template <int G, bool RECUR>
__device__ void improve_max_group(const double* __restrict__ y, int n, double ix0, int depth, int nz_lo, int nz_hi,
                                  int lg, bool live, double& xm, double& ym) {
    const double SQRT_EPS = 1.4901161193847656e-08, TOL3 = 1e-10 / 3.0;
    double a = ix0 + 1.0 - 1.0, b = ix0 + 1.0 + 1.0;     // 1-based bracket
    double v = a + GOLD * (b - a);
    double fv = -sinc_group<G, RECUR>(y, n, v - 1.0, depth, nz_lo, nz_hi, lg);
    double x = v, w = v, fx = fv, fw = fv;
    bool active = live;
    for (int it = 0; it < 60; ++it) {
        const double rng = b - a, mid = 0.5 * (a + b);
        const double tol_act = SQRT_EPS * fabs(x) + TOL3;
        if (fabs(x - mid) + 0.5 * rng <= 2.0 * tol_act) active = false;
        if (!__any(active)) break;
        double step = GOLD * (x < mid ? b - x : a - x);
        if (fabs(x - w) >= tol_act) {
            const double t = (x - w) * (fx - fv);
            double q = (x - v) * (fx - fw);
            double p = (x - v) * q - (x - w) * t;
            q = 2.0 * (q - t);
            if (q > 0.0) p = -p; else q = -q;
            if (fabs(p) < fabs(step * q) && p > q * (a - x + 2.0 * tol_act) && p < q * (b - x - 2.0 * tol_act))
                step = p / q;
        }
        if (fabs(step) < tol_act) step = step > 0.0 ? tol_act : -tol_act;
        const double tt = x + step;
        const double ft = -sinc_group<G, RECUR>(y, n, tt - 1.0, depth, nz_lo, nz_hi, lg);
        if (active) {
            if (ft <= fx) {
                if (tt < x) b = x; else a = x;
                v = w; w = x; x = tt;
                fv = fw; fw = fx; fx = ft;
            } else {
                if (tt < x) a = tt; else b = tt;
                if (ft <= fw || w == x) { v = w; w = tt; fv = fw; fw = ft; }
                else if (ft <= fv || v == x || v == w) { v = tt; fv = ft; }
            }
        }
    }
    xm = x - 1.0;
    ym = -fx;
}

// ---- pitch candidates per frame (AC: Hanning-windowed autocorrelation; CC: forward cross-correlation) --
// dynamic LDS: seg[seg_len] doubles, r[2*brent_ixmax+1] doubles
struct FrameOut {     // per frame, written contiguously: intensity, ncand, freq[MAXC], strength[MAXC]
    double intensity;
    double ncand;
    double freq[MAXC];
    double strength[MAXC];
};

typedef double double4_t __attribute__((ext_vector_type(4)));
// stored half-width of the candidate kernel's correlation array (see pitch_cand_kernel)
// [lo, hi] = the lags the candidate kernel can touch: the candidates' lags widened by the interpolation depth, never
// beyond 2 L + 2 (zeros past the non-zero range |lag| <= L that a tap loop may still multiply) nor the array itself
__host__ __device__ inline void pitch_r_range(int rc, int L, int min_lag, int max_lag, int depth, int* lo, int* hi) {
    const int w = 2 * L + 2, span = w < rc ? w : rc;
    const int lag_lo = min_lag > 2 ? min_lag : 2;
    int lag_hi = max_lag - 1;
    if (lag_hi > rc - 1) lag_hi = rc - 1;
    const int d = depth > 30 ? depth : 30;
    int a = lag_lo - d - 3, b = lag_hi + d + 3;
    *lo = a < -span ? -span : a;
    *hi = b > span ? span : b;
}
// ---- sinc interpolation as a polynomial in the fractional position -----------------------------------------
// Between two samples the depth-d interpolation is S(b + frac) = sum_o W_o(frac) y[b + o], o = -(d-1) .. d, and
// every weight W_o is a smooth function of frac in [0, 1] alone (as long as the depth is not clipped by the
// array ends).  With the degree-15 Chebyshev coefficients of the weights tabulated once on the host
// (cheb[o + d - 1][j]), the 16 coefficients of S on a cell cost one pass over the taps, after which every Brent
// evaluation is a 16-term Clenshaw recurrence instead of 2d reciprocal-and-cosine terms.  The fit error
// (< 1e-12) is below the rounding of the direct formula near integer positions.
constexpr int NCH = 16;

__device__ __forceinline__ double cheb_eval(const double* __restrict__ c, double frac) {
    const double t = 2.0 * frac - 1.0, t2 = 2.0 * t;
    double b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int j = NCH - 1; j >= 1; --j) { const double b0 = fma(t2, b1, c[j] - b2); b2 = b1; b1 = b0; }
    return fma(t, b1, c[0] - b2);
}

// Praat NUMimproveMaximum (sinc) on the two cells around the 0-based integer position x0; P = [2][NCH] coefficients
// (cell 0 = [x0-1, x0], cell 1 = [x0, x0+1]).  One lane per candidate; the loop runs while any lane is active.
// The 2 x 16 coefficients of the lane's candidate stay in registers for the whole search: an evaluation is then a
// per-coefficient select and ONE dependent multiply-add per Clenshaw step (the subtraction c_j - b_{j+2} does not wait
// for b_{j+1}); reading the cell's row from LDS per evaluation put a memory round trip in front of every chain.
__device__ void improve_max_cheb(const double* __restrict__ Pc, int x0, bool live, double& xm, double& ym) {
    const double SQRT_EPS = 1.4901161193847656e-08, TOL3 = 1e-10 / 3.0;
    const double ix1 = (double)x0 + 1.0;                    // 1-based like Praat
    double c0[NCH], c1[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { c0[j] = Pc[j]; c1[j] = Pc[NCH + j]; }
    auto f = [&](double v1) {                               // v1: 1-based position in [ix1-1, ix1+1]
        double fl = floor(v1);
        double cell = fl - (ix1 - 1.0);
        cell = cell < 0.0 ? 0.0 : (cell > 1.0 ? 1.0 : cell);
        const double frac = v1 - (ix1 - 1.0 + cell);
        const bool hi = cell > 0.5;
        const double t = 2.0 * frac - 1.0, t2 = 2.0 * t;
        double b1 = 0.0, b2 = 0.0;
#pragma unroll
        for (int j = NCH - 1; j >= 1; --j) { const double b0 = fma(t2, b1, (hi ? c1[j] : c0[j]) - b2); b2 = b1; b1 = b0; }
        return -fma(t, b1, (hi ? c1[0] : c0[0]) - b2);
    };
    double a = ix1 - 1.0, b = ix1 + 1.0;
    double v = a + GOLD * (b - a);
    double fv = f(v);
    double x = v, w = v, fx = fv, fw = fv;
    bool active = live;
    for (int it = 0; it < 60; ++it) {
        const double rng = b - a, mid = 0.5 * (a + b);
        const double tol_act = SQRT_EPS * fabs(x) + TOL3;
        if (fabs(x - mid) + 0.5 * rng <= 2.0 * tol_act) active = false;
        if (!__any(active)) break;
        double step = GOLD * (x < mid ? b - x : a - x);
        if (fabs(x - w) >= tol_act) {
            const double t = (x - w) * (fx - fv);
            double q = (x - v) * (fx - fw);
            double p = (x - v) * q - (x - w) * t;
            q = 2.0 * (q - t);
            if (q > 0.0) p = -p; else q = -q;
            if (fabs(p) < fabs(step * q) && p > q * (a - x + 2.0 * tol_act) && p < q * (b - x - 2.0 * tol_act))
                step = p / q;
        }
        if (fabs(step) < tol_act) step = step > 0.0 ? tol_act : -tol_act;
        const double tt = x + step;
        const double ft = f(tt);
        if (active) {
            if (ft <= fx) {
                if (tt < x) b = x; else a = x;
                v = w; w = x; x = tt;
                fv = fw; fw = fx; fx = ft;
            } else {
                if (tt < x) a = tt; else b = tt;
                if (ft <= fw || w == x) { v = w; w = tt; fv = fw; fw = ft; }
                else if (ft <= fv || v == x || v == w) { v = tt; fv = ft; }
            }
        }
    }
    xm = x - 1.0;
    ym = -fx;
}

struct RefineArgs {
    const double* r; int RN, RC, depth, nz_lo, nz_hi, ncand; double margin;
    const int* place; double* cf; double* cs;
};
// refine every kept candidate: maximise the sinc-interpolated correlation, 256/G candidates per round.
// With every path cost zero (harmonicity pass) a candidate far below the best first-pass strength could
// be left unrefined (margin > 0); that is off by default because it moved a few frames' selection.
template <int G, bool RECUR>
__device__ void refine_candidates(const RefineArgs& A, int tid, int nthreads) {
    const int lane = tid & 63, lg = lane & (G - 1), gidx = (tid >> 6) * (64 / G) + lane / G;
    double best_first = 0.0;
    for (int k = 1; k < A.ncand; ++k) best_first = fmax(best_first, A.cs[k]);
    __syncthreads();
    for (int kb = 1; kb < A.ncand; kb += nthreads / G) {
        const int k = kb + gidx;
        const bool live = k < A.ncand && (A.margin <= 0.0 || A.cs[k < A.ncand ? k : 1] >= best_first - A.margin);
        double xm, ym;
        improve_max_group<G, RECUR>(A.r, A.RN, (double)(A.place[live ? k : 1] + A.RC), A.depth, A.nz_lo, A.nz_hi, lg,
                                    live, xm, ym);
        if (ym > 1.0) ym = 1.0 / ym;
        if (live && lg == 0) { A.cf[k] = 1.0 / DXS / (xm - A.RC); A.cs[k] = ym; }
    }
}

// ---- AC: windowed autocorrelation by FFT ---------------------------------------------------------------------
// Praat's Sound_to_Pitch (ac) transforms the windowed frame with an FFT of nsampFFT >= 1.5 nsamp_window points, squares
// the spectrum and transforms back; so does this kernel, in fp64: about 2.5 N log2 N flops per transform against
// 2 nw L for the direct sum (nine times fewer at nw = 960, L = 512).  The real transform of N points is a complex
// transform of M = N / 2 points on z[j] = x[2 j] + i x[2 j + 1] (the zero-padded frame, as it lies in LDS, IS z), a
// pass that separates X[k], squares it and packs the even spectrum P back into M complex points, and a second complex
// transform whose output is r[2 j] + i r[2 j + 1].  The complex transform is a Stockham autosort FFT (radix 4, a final
// radix 2 when M is not a power of 4): natural order in and out, ping-pong between two LDS buffers, 256 threads.
// Twiddles W_N^k = exp(-2 pi i k / N), k < N / 2, come from a table built on the host in double precision.
typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double2_t cmul(double2_t a, double2_t b) {
    return double2_t{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ double2_t tw_at(const double2_t* __restrict__ tw, int t, int M) {   // W_N^t, 0 <= t < N
    const bool neg = t >= M;
    const double2_t w = tw[neg ? t - M : t];
    return neg ? double2_t{-w.x, -w.y} : w;
}

// The transform length is a template parameter: the stages unroll, and for M <= 1024 every twiddle a thread needs
// (a function of the stage and the thread only) is fetched ONCE, in front of the frame's first pass over the samples, and
// kept in registers; the table loads (L2 latency) no longer sit between the barriers of every stage.
template <int LOG2M>
struct FftPlan {
    static constexpr int M = 1 << LOG2M, N = 2 * M;
    static constexpr int N4 = LOG2M / 2;                    // radix-4 stages
    static constexpr bool HAS2 = (LOG2M & 1) != 0;          // one radix-2 stage behind them
    static constexpr int T4 = M / 4, T2 = M / 2;
    static constexpr int BPT4 = T4 > 256 ? T4 / 256 : 1, BPT2 = T2 > 256 ? T2 / 256 : 1;
    static constexpr bool PRE = LOG2M <= 11;                // twiddles in registers
    static constexpr int NTW4 = PRE && N4 > 1 ? (N4 - 1) * BPT4 * 3 : 1, NTW2 = PRE && HAS2 ? BPT2 : 1;
};

template <int LOG2M>
struct FftTw {
    double2_t w4[FftPlan<LOG2M>::NTW4];
    double2_t w2[FftPlan<LOG2M>::NTW2];
};

template <int LOG2M>
__device__ __forceinline__ void fft_load_twiddles(FftTw<LOG2M>& R, const double2_t* __restrict__ tw, int tid) {
    using PL = FftPlan<LOG2M>;
    if (!PL::PRE) return;
#pragma unroll
    for (int st = 1; st < PL::N4; ++st) {
        const int Ns = 1 << (2 * st), step = PL::N / (Ns * 4);
#pragma unroll
        for (int bq = 0; bq < PL::BPT4; ++bq) {
            const int j = tid + 256 * bq, t1 = (j & (Ns - 1)) * step;
#pragma unroll
            for (int r = 1; r <= 3; ++r) R.w4[((st - 1) * PL::BPT4 + bq) * 3 + r - 1] = tw_at(tw, (r * t1) & (PL::N - 1), PL::M);
        }
    }
    if (PL::HAS2) {
        const int Ns = 1 << (2 * PL::N4), step = PL::N / (Ns * 2);
#pragma unroll
        for (int bq = 0; bq < PL::BPT2; ++bq) R.w2[bq] = tw_at(tw, ((tid + 256 * bq) & (Ns - 1)) * step, PL::M);
    }
}

// Between the first two radix-4 stages element e lives at slot fsw(e) (low two bits XOR-ed with bits 3-4): stage 0 writes
// elements 4 j + r from lane j, a 64-byte lane stride = 4-way bank conflict for the 16-byte stores of 8 consecutive lanes
// (PMC: 30 % of the LDS-active cycles of the autocorrelation kernel were conflict cycles, the LDS busy 65 % of the time);
// swizzled, the 8 lanes hit 8 different slots of the 128-byte bank row, and the unit-stride reads of stage 1 stay
// conflict-free (the permutation stays inside aligned blocks of 4 slots).  Every other pass sees the natural order.
__device__ __forceinline__ int fsw(int e) { return e ^ ((e >> 3) & 3); }

// forward complex FFT of M points from `a` (result in the returned buffer, `a` or `b`); every stage ends at a barrier
template <int LOG2M>
__device__ __forceinline__ double2_t* fft_stockham(double2_t* a, double2_t* b, const FftTw<LOG2M>& R,
                                                   const double2_t* __restrict__ tw, int tid) {
    using PL = FftPlan<LOG2M>;
    constexpr int M = PL::M, N = PL::N, T4 = PL::T4, T2 = PL::T2;
    double2_t* src = a;
    double2_t* dst = b;
#pragma unroll
    for (int st = 0; st < PL::N4; ++st) {
        const int Ns = 1 << (2 * st), step = N / (Ns * 4);
#pragma unroll
        for (int bq = 0; bq < PL::BPT4; ++bq) {
            const int j = tid + 256 * bq;
            if (T4 >= 256 || j < T4) {
                const int k = j & (Ns - 1);
                const bool rs = st == 1;                                  // stage 0 stored swizzled
                double2_t v0 = src[rs ? fsw(j) : j], v1 = src[rs ? fsw(j + T4) : j + T4], v2 = src[rs ? fsw(j + 2 * T4) : j + 2 * T4],
                          v3 = src[rs ? fsw(j + 3 * T4) : j + 3 * T4];
                if (st > 0) {
                    if (PL::PRE) {
                        const int o = ((st - 1) * PL::BPT4 + bq) * 3;
                        v1 = cmul(v1, R.w4[o]);
                        v2 = cmul(v2, R.w4[o + 1]);
                        v3 = cmul(v3, R.w4[o + 2]);
                    } else {
                        const int t1 = k * step;
                        v1 = cmul(v1, tw_at(tw, t1, M));
                        v2 = cmul(v2, tw_at(tw, 2 * t1, M));
                        v3 = cmul(v3, tw_at(tw, 3 * t1, M));
                    }
                }
                const double2_t a0 = v0 + v2, a1 = v0 - v2, a2 = v1 + v3;
                const double2_t d = v1 - v3;
                const double2_t a3 = double2_t{d.y, -d.x};             // (v1 - v3) * (-i)
                const int j0 = ((j - k) << 2) + k;
                const bool ws = st == 0 && PL::N4 >= 2;
                dst[ws ? fsw(j0) : j0] = a0 + a2;
                dst[ws ? fsw(j0 + Ns) : j0 + Ns] = a1 + a3;
                dst[ws ? fsw(j0 + 2 * Ns) : j0 + 2 * Ns] = a0 - a2;
                dst[ws ? fsw(j0 + 3 * Ns) : j0 + 3 * Ns] = a1 - a3;
            }
        }
        __syncthreads();
        double2_t* t_ = src; src = dst; dst = t_;
    }
    if (PL::HAS2) {                                                     // one radix-2 stage left (M = 2 * 4^a)
        const int Ns = 1 << (2 * PL::N4), step = N / (Ns * 2);
#pragma unroll
        for (int bq = 0; bq < PL::BPT2; ++bq) {
            const int j = tid + 256 * bq;
            if (T2 >= 256 || j < T2) {
                const int k = j & (Ns - 1);
                const double2_t v0 = src[j];
                const double2_t v1 = cmul(src[j + T2], PL::PRE ? R.w2[bq] : tw_at(tw, k * step, M));
                const int j0 = ((j - k) << 1) + k;
                dst[j0] = v0 + v1;
                dst[j0 + Ns] = v0 - v1;
            }
        }
        __syncthreads();
        double2_t* t_ = src; src = dst; dst = t_;
    }
    return src;
}

constexpr int AC_FRAMES_PER_WG = 16;

template <int LOG2M>
__global__ __launch_bounds__(256) void pitch_ac_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                       const double* __restrict__ gpeak, const double* __restrict__ win,
                                                       const double* __restrict__ wr, const PitchParams P,
                                                       const double2_t* __restrict__ tw, double* __restrict__ rbuf,
                                                       int rstride, int max_frames) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const ClipInfo c = ci[blockIdx.y];
    if ((int)blockIdx.x * AC_FRAMES_PER_WG >= c.n_frames) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int N = FftPlan<LOG2M>::N, M = FftPlan<LOG2M>::M;
    const int nw = P.nsamp_window, L = P.brent_ixmax;
    FftTw<LOG2M> twr;
    fft_load_twiddles<LOG2M>(twr, tw, tid);                // in flight during the passes over the samples
    constexpr int NPK = (M / 2) / 256 + 1;                  // spectrum pass: k = tid + 256 i <= M / 2
    double2_t twp[NPK];
#pragma unroll
    for (int i = 0; i < NPK; ++i) twp[i] = tw[tid + 256 * i <= M / 2 ? tid + 256 * i : 0];
    double* seg = reinterpret_cast<double*>(smem_raw);      // [N]: the windowed frame, zero-padded = M complex points
    double* buf = seg + N;                                  // [N]
    double* s_red = buf + N;                                // [4]
    double* s_val = s_red + 4;                              // [4]
    const float* x = wav + c.sample_off;
    const int n = c.n_samples;
    const double gp = gpeak[blockIdx.y];
    // a workgroup takes AC_FRAMES_PER_WG consecutive frames: the twiddles (72 KB of table reads per workgroup at M = 1024)
    // are fetched once for all of them
    for (int f = blockIdx.x * AC_FRAMES_PER_WG; f < (int)(blockIdx.x + 1) * AC_FRAMES_PER_WG && f < c.n_frames; ++f) {
    double* rb = rbuf + ((int64_t)blockIdx.y * max_frames + f) * rstride;   // r[0..L], then the intensity
    const double t = c.t1 + f * P.dt;
    const int64_t left = low_index(t, c.x1), right = left + 1;
    // local mean over one longest period to each side (divisor 2*nsamp_period as in Praat)
    {
        int64_t s0 = right - P.nsamp_period, s1 = left + P.nsamp_period;
        s0 = s0 < 0 ? 0 : (s0 > n - 1 ? n - 1 : s0);
        s1 = s1 < 0 ? 0 : (s1 > n - 1 ? n - 1 : s1);
        double s = 0.0;
        for (int64_t i = s0 + tid; i <= s1; i += 256) s += (double)x[i];
        s = group_sum<64>(s);
        if (lane == 0) s_red[wv] = s;
    }
    __syncthreads();
    const double local_mean = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (2.0 * P.nsamp_period);
    const int64_t start = right - P.half_window;
    if (start >= 0 && start + nw <= n) {                    // the window lies inside the sound: no index clamps
        const float* xs = x + start;
#pragma unroll 4
        for (int j = tid; j < N; j += 256) seg[j] = j < nw ? ((double)xs[j] - local_mean) * win[j] : 0.0;
    } else {
        for (int j = tid; j < N; j += 256) {
            int64_t i = start + j;
            i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
            seg[j] = j < nw ? ((double)x[i] - local_mean) * win[j] : 0.0;
        }
    }
    __syncthreads();
    // local peak over half a longest period around the window centre
    {
        int a = P.half_window - P.half_period, b = P.half_window + P.half_period;
        a = a < 0 ? 0 : a;
        b = b > nw ? nw : b;
        double m = 0.0;
        for (int j = a + tid; j < b; j += 256) m = fmax(m, fabs(seg[j]));
        m = wave_max_dpp(m);
        if (lane == 0) s_val[wv] = m;
    }
    __syncthreads();
    const double local_peak = fmax(fmax(s_val[0], s_val[1]), fmax(s_val[2], s_val[3]));
    const double intensity = gp > 0.0 ? (local_peak > gp ? 1.0 : local_peak / gp) : 0.0;
    if (P.debug_stop == 1) { __syncthreads(); continue; }

    double2_t* za = reinterpret_cast<double2_t*>(seg);
    double2_t* zb = reinterpret_cast<double2_t*>(buf);
    double2_t* Z = fft_stockham<LOG2M>(za, zb, twr, tw, tid);
    double2_t* Y = Z == za ? zb : za;
    // X[k] = E + W^k O, X[M - k] = conj(E - W^k O) with E = (Z[k] + conj Z[M-k]) / 2, O = (Z[k] - conj Z[M-k]) / 2i;
    // P = |X|^2 is real and even, and the M-point input of the transform back is
    // Y[k] = (P[k] + P[M-k]) + i conj(W^k) (P[k] - P[M-k]); it is stored conjugated (inverse by the forward transform)
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
        const int k = tid + 256 * i;
        if (k > M / 2) continue;
        if (k == 0) {
            const double2_t z0 = Z[0];
            const double p0 = (z0.x + z0.y) * (z0.x + z0.y), pm = (z0.x - z0.y) * (z0.x - z0.y);
            Y[0] = double2_t{p0 + pm, -(p0 - pm)};
        } else if (k == M / 2) {
            const double2_t zk = Z[k];
            Y[k] = double2_t{2.0 * (zk.x * zk.x + zk.y * zk.y), 0.0};
        } else {
            const double2_t zk = Z[k], zm = Z[M - k];
            const double2_t E = double2_t{0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y)};
            const double2_t D = double2_t{0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y)};
            const double2_t O = double2_t{D.y, -D.x};
            const double2_t w = twp[i];
            const double2_t T = cmul(w, O);
            const double2_t xa = E + T, xb = E - T;
            const double pk = xa.x * xa.x + xa.y * xa.y, pm = xb.x * xb.x + xb.y * xb.y;
            const double sum = pk + pm, d = pk - pm;
            Y[k] = double2_t{sum + w.y * d, -(w.x * d)};
            Y[M - k] = double2_t{sum - w.y * d, -(w.x * d)};
        }
    }
    __syncthreads();
    const double* r = reinterpret_cast<const double*>(fft_stockham<LOG2M>(Y, Y == za ? zb : za, twr, tw, tid));
    if (P.debug_stop == 2) continue;
    // r[2 j] = Re, r[2 j + 1] = -Im of the (conjugated) output; normalise into the global row
    if (tid == 0) { rb[0] = 1.0; rb[L + 1] = intensity; }
    const double r0 = r[0];
    for (int l = 1 + tid; l <= L; l += 256) {
        const double v = (l & 1) ? -r[l] : r[l];
        rb[l] = r0 > 0.0 ? v / (r0 * wr[l]) : 0.0;
    }
    __syncthreads();                                        // the next frame overwrites both buffers
    }
}

// ---- CC: forward cross-correlation by FFT ----------------------------------------------------------------------
// r(l) = sum_{j < nw} seg[j] seg[j + l], l = 0 .. L, is the linear cross-correlation of a = seg[0, nw) with b = seg[0, nw + L]:
// both are real, so ONE complex FFT of N >= nw + L + 1 points on z = a + i b gives A[k] = (Z[k] + conj Z[N-k]) / 2 and
// B[k] = (Z[k] - conj Z[N-k]) / 2i; C = conj(A) B is the spectrum of the correlation, Hermitian, and goes back through a
// complex FFT of N / 2 points like the autocorrelation kernel's second transform.  (On the fp64 matrix pipe the direct sum
// cost 0.6 M multiply-adds per frame at the 60 Hz floor; this is 0.17 M flops.)  Normalisation as before: r / sqrt(sumx2 sumy2(l))
// with sumy2 from a block prefix sum of the squares, taken before the transforms reuse the buffers.
constexpr int CC_FRAMES_PER_WG = 16;

template <int LOG2N>
__global__ __launch_bounds__(256) void pitch_cc_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                       const double* __restrict__ gpeak, const PitchParams P,
                                                       const double2_t* __restrict__ tw1, const double2_t* __restrict__ tw2,
                                                       double* __restrict__ rbuf, int rstride, int max_frames) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const ClipInfo c = ci[blockIdx.y];
    if ((int)blockIdx.x * CC_FRAMES_PER_WG >= c.n_frames) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int N = 1 << LOG2N, M = N >> 1;               // complex transform lengths: N forward, M back
    const int nw = P.nsamp_window, L = P.max_lag;
    const int seg_len = nw + L + 1;
    FftTw<LOG2N> twa;                                       // tw1 = W_2N^k (k < N): twiddles of the N-point transform
    FftTw<LOG2N - 1> twb;                                   // tw2 = W_N^k (k < M): the M-point transform and the spectrum pass
    fft_load_twiddles<LOG2N>(twa, tw1, tid);
    fft_load_twiddles<LOG2N - 1>(twb, tw2, tid);
    constexpr int NPK = (M / 2) / 256 + 1;                  // spectrum pass: k = tid + 256 i <= M / 2
    double2_t twp[NPK];
#pragma unroll
    for (int i = 0; i < NPK; ++i) twp[i] = tw2[tid + 256 * i <= M / 2 ? tid + 256 * i : 0];
    double2_t* za = reinterpret_cast<double2_t*>(smem_raw);  // [N]
    double2_t* zb = za + N;                                  // [N]
    double* s_sy = reinterpret_cast<double*>(zb + N);        // [L + 1] sumy2(l)
    double* s_red = s_sy + ((L + 2) & ~1);                   // [8]
    double* s_val = s_red + 8;                               // [4]
    double* s_scan = s_val + 4;                              // [16] wave totals of the four 256-lag chunks of the scan
    const float* x = wav + c.sample_off;
    const int n = c.n_samples;
    const double gp = gpeak[blockIdx.y];
    for (int f = blockIdx.x * CC_FRAMES_PER_WG; f < (int)(blockIdx.x + 1) * CC_FRAMES_PER_WG && f < c.n_frames; ++f) {
    double* rb = rbuf + ((int64_t)blockIdx.y * max_frames + f) * rstride;   // r[0..L], then the intensity
    const double t = c.t1 + f * P.dt;
    const int64_t left = low_index(t, c.x1), right = left + 1;
    // local mean over one longest period to each side (divisor 2*nsamp_period as in Praat)
    {
        int64_t s0 = right - P.nsamp_period, s1 = left + P.nsamp_period;
        s0 = s0 < 0 ? 0 : (s0 > n - 1 ? n - 1 : s0);
        s1 = s1 < 0 ? 0 : (s1 > n - 1 ? n - 1 : s1);
        double s = 0.0;
        for (int64_t i = s0 + tid; i <= s1; i += 256) s += (double)x[i];
        s = group_sum<64>(s);
        if (lane == 0) s_red[wv] = s;
    }
    __syncthreads();
    const double local_mean = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (2.0 * P.nsamp_period);
    int loc_max_lag;
    {
        // Praat: startTime = t - 0.5 * (1 / minimumPitch + dt_window), dt_window = periods / minimumPitch
        const double start_time = t - 0.5 * (1.0 / P.min_pitch + P.dt_window);
        int64_t start = low_index(start_time, c.x1);
        if (start < 0) start = 0;
        int64_t span = L + nw;
        if (span > n - start) span = n - start;
        loc_max_lag = (int)(span - nw);
        // one pass: z = a + i b, the local peak (|b| over half a longest period around the window centre) and sumx2 = sum a^2
        int pa = P.half_window - P.half_period, pb = P.half_window + P.half_period;
        pa = pa < 0 ? 0 : pa;
        pb = pb > nw ? nw : pb;
        double m = 0.0, sx = 0.0;
        const float* xs = x + start;                          // start >= 0
        const int avail = (int)((n - start) < (int64_t)seg_len ? (n - start) : (int64_t)seg_len);   // samples of seg inside the sound
#pragma unroll 4
        for (int j = tid; j < N; j += 256) {
            const double v = j < avail ? ((double)xs[j] - local_mean) : 0.0;
            za[j] = double2_t{j < nw ? v : 0.0, v};          // z = a + i b
            if (j >= pa && j < pb) m = fmax(m, fabs(v));
            if (j < nw) sx += v * v;
        }
        m = wave_max_dpp(m);
        sx = group_sum<64>(sx);
        if (lane == 0) { s_val[wv] = m; s_red[4 + wv] = sx; }
    }
    __syncthreads();
    const double local_peak = fmax(fmax(s_val[0], s_val[1]), fmax(s_val[2], s_val[3]));
    const double intensity = gp > 0.0 ? (local_peak > gp ? 1.0 : local_peak / gp) : 0.0;
    const double sumx2 = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    // sumy2(l) = sum_{j=l}^{l+nw-1} b_j^2 = sumx2 + sum_{i<l} (b_{i+nw}^2 - b_i^2): an inclusive scan over the L lags
    {
        constexpr int NQ = 4;                                 // L <= 1023
        double inc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = tid + 256 * q;                      // term i feeds sumy2(i + 1)
            double d = 0.0;
            if (i < L) { const double u = za[i + nw].y, w0 = za[i].y; d = u * u - w0 * w0; }
            double sc = d;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const double t2 = __shfl_up(sc, o, 64); if (lane >= o) sc += t2; }
            inc[q] = sc;
            if (lane == 63) s_scan[4 * q + wv] = sc;
        }
        __syncthreads();
        if (tid == 0) s_sy[0] = sumx2;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = tid + 256 * q;
            if (i < L) {
                double off = 0.0;
                for (int z = 0; z < 4 * q + wv; ++z) off += s_scan[z];
                s_sy[i + 1] = sumx2 + (off + inc[q]);
            }
        }
    }
    if (P.debug_stop == 1) continue;

    double2_t* Z = fft_stockham<LOG2N>(za, zb, twa, tw1, tid);
    double2_t* Y = Z == za ? zb : za;
    // C[k] = conj(A[k]) B[k]; Y[k] = (C[k] + conj C[M-k]) + i conj(W_N^k) (C[k] - conj C[M-k]), stored conjugated
    auto spec = [&](int k) {                                 // C[k] for 0 <= k <= M
        const double2_t zk = Z[k], zn = Z[(N - k) & (N - 1)];
        const double2_t A = double2_t{0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y)};
        const double2_t Bm = double2_t{0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x)};   // (Z[k] - conj Z[N-k]) / 2i
        return double2_t{A.x * Bm.x + A.y * Bm.y, A.x * Bm.y - A.y * Bm.x};          // conj(A) * B
    };
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
        const int k = tid + 256 * i;
        if (k > M / 2) continue;
        const double2_t ck = spec(k), cm = spec(M - k);
        const double2_t w = twp[i];                          // W_N^k
        {   // Y[k]
            const double2_t su = double2_t{ck.x + cm.x, ck.y - cm.y}, di = double2_t{ck.x - cm.x, ck.y + cm.y};
            // i conj(w) di = i (w.x - i w.y)(di.x + i di.y) = (w.y di.x - w.x di.y) ... real: -(w.x di.y - w.y di.x)?  expand:
            // conj(w) di = (w.x di.x + w.y di.y) + i (w.x di.y - w.y di.x);  times i: -(w.x di.y - w.y di.x) + i (w.x di.x + w.y di.y)
            const double2_t yy = double2_t{su.x - (w.x * di.y - w.y * di.x), su.y + (w.x * di.x + w.y * di.y)};
            Y[k] = double2_t{yy.x, -yy.y};
        }
        if (k != 0 && k != M - k) {   // Y[M - k]: roles swapped, W_N^(M-k) = -conj(W_N^k), so conj(W_N^(M-k)) = -w
            const double2_t su = double2_t{cm.x + ck.x, cm.y - ck.y}, di = double2_t{cm.x - ck.x, cm.y + ck.y};
            // i * (-w) * di = -i (w.x + i w.y)(di.x + i di.y) = (w.x di.y + w.y di.x) - i (w.x di.x - w.y di.y)
            const double2_t yy = double2_t{su.x + (w.x * di.y + w.y * di.x), su.y - (w.x * di.x - w.y * di.y)};
            Y[M - k] = double2_t{yy.x, -yy.y};
        }
    }
    __syncthreads();
    const double* r = reinterpret_cast<const double*>(fft_stockham<LOG2N - 1>(Y, Y == za ? zb : za, twb, tw2, tid));
    if (P.debug_stop == 2) continue;
    // r[2 j] = Re, r[2 j + 1] = -Im of the (conjugated) output, times N
    if (tid == 0) { rb[0] = 1.0; rb[L + 1] = intensity; }
    const double inv_n = 1.0 / (double)N;
    for (int l = 1 + tid; l <= L; l += 256) {
        const double v = ((l & 1) ? -r[l] : r[l]) * inv_n;
        const double den = sumx2 * s_sy[l];
        rb[l] = (l <= loc_max_lag && den > 0.0) ? v / sqrt(den) : 0.0;
    }
    __syncthreads();                                        // the next frame overwrites the buffers
    }
}

// ---- the two correlation kernels with ONE WAVE per frame (csrc/wave_fft.h) --------------------------------------
// Transform lengths up to 2048 complex points (every analysis of the MSHDS feature scripts at 16 kHz; shorter transforms
// are zero-padded up to 512 / 1024 points, which returns the same linear correlation) run here: the frame lives in the registers of one wavefront from the sample loads to the normalised correlation row.  The local
// mean, the window, the local peak and sum x^2 are taken on the registers the transform starts from (wave reductions by
// DPP, no LDS round trip, no workgroup barrier anywhere), the spectrum step evaluates every conjugate pair once, and only
// the lags the candidate kernel reads are normalised and stored.  A wave takes WF_FRAMES consecutive frames so that its
// five base twiddles are fetched once.
constexpr int WF_FRAMES = 8;

__device__ __forceinline__ wfft::cplx ld_tw(const double2_t* __restrict__ tw, int i) {
    const double2_t w = tw[i];
    return wfft::cplx{w.x, w.y};
}
// 1 / sqrt(d) for d > 0: hardware estimate + two Newton steps
__device__ __forceinline__ double fast_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    y = y * (1.5 - 0.5 * d * y * y);
    return y;
}
// inclusive prefix sum over the 64 lanes: four row_shr steps inside the rows of 16, the row totals through v_readlane
__device__ __forceinline__ double wave_scan_incl(double v, int lane) {
    v += dpp_f64<0x111>(v);
    v += dpp_f64<0x112>(v);
    v += dpp_f64<0x114>(v);
    v += dpp_f64<0x118>(v);
    const double t0 = readlane_f64(v, 15), t1 = readlane_f64(v, 31), t2 = readlane_f64(v, 47);
    const int row = lane >> 4;
    return v + (row == 0 ? 0.0 : (row == 1 ? t0 : (row == 2 ? t0 + t1 : (t0 + t1) + t2)));
}

template <int R>
__global__ __launch_bounds__(64, R == 32 ? 2 : (R == 16 ? 3 : 4)) void pitch_ac_wave_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                           const double* __restrict__ gpeak, const double* __restrict__ win,
                                                           const double* __restrict__ wr, const PitchParams P,
                                                           const double2_t* __restrict__ tw, double* __restrict__ rbuf,
                                                           int rstride, int max_frames) {
    using namespace wfft;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const ClipInfo c = ci[blockIdx.y];
    const int f0 = blockIdx.x * WF_FRAMES;
    if (f0 >= c.n_frames) return;
    const int lane_ = threadIdx.x;
    constexpr int S = 64 * R, N = 2 * S, L2 = 64 / R;        // tw = W_N^k, k < S
    const int nw = P.nsamp_window, L = P.brent_ixmax;
    const float* x = wav + c.sample_off;
    const int n = c.n_samples;
    const double gp = gpeak[blockIdx.y];
    LdsMem mem{lds};
    const int f1 = f0 + WF_FRAMES < c.n_frames ? f0 + WF_FRAMES : c.n_frames;
    int pa = P.half_window - P.half_period, pb = P.half_window + P.half_period;
    pa = pa < 0 ? 0 : pa;
    pb = pb > nw ? nw : pb;
#pragma unroll 1
    for (int f = f0; f < f1; ++f) {
        // Everything below that depends only on the lane (addresses, range predicates, the powers of the base twiddles) is
        // loop-invariant, and hoisted out of the frame loop it costs more registers than the frame itself: the lane index and
        // the base twiddles pass through an empty asm so that they count as redefined per frame.
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        // base twiddles (three 16-byte loads per frame, L1): W_S^lane = W_N^(2 lane), W_64^(lane % L2), W_N^lane
        cplx w_s = ld_tw(tw, 2 * lane), w_b = ld_tw(tw, (lane % L2) * (N / 64));
        double* rb = rbuf + ((int64_t)blockIdx.y * max_frames + f) * rstride;   // r[0..L], then the intensity
        const double t = c.t1 + f * P.dt;
        const int left = (int)low_index(t, c.x1), right = left + 1;
        // The raw samples first, as the S complex points z[k] = x[2 k] + i x[2 k + 1], k = lane + 64 m: every load of the frame is
        // in flight at once.  (A separate pass for the local mean in front of them made the frame wait for memory twice.)
        const int start = right - P.half_window;
        const bool inside = start >= 0 && start + nw <= n;        // the window lies inside the sound
        // (Loads are unconditional, on clamped indices, and what lies outside the window is zeroed afterwards: a load inside a
        // branch is waited for inside that branch, and the frame would cross the memory latency once per register.)
        cplx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int j = 128 * m + 2 * lane;
            int i0 = start + j, i1 = i0 + 1;
            i0 = i0 < 0 ? 0 : (i0 > n - 1 ? n - 1 : i0);
            i1 = i1 < 0 ? 0 : (i1 > n - 1 ? n - 1 : i1);
            v[m] = cplx{(double)x[i0], (double)x[i1]};
        }
        // local mean over one longest period to each side (divisor 2*nsamp_period as in Praat): from the registers when the
        // window holds that range (always, for windows of two periods and more away from the ends of the sound)
        double local_mean;
        {
            int s0 = right - P.nsamp_period, s1 = left + P.nsamp_period;
            s0 = s0 < 0 ? 0 : (s0 > n - 1 ? n - 1 : s0);
            s1 = s1 < 0 ? 0 : (s1 > n - 1 ? n - 1 : s1);
            double sm = 0.0;
            if (inside && s0 >= start && s1 < start + nw) {
                const int a0 = s0 - start, a1 = s1 - start;
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    const int jb = 128 * m, j = jb + 2 * lane;
                    if (jb + 127 >= a0 && jb <= a1) {
                        if (j >= a0 && j <= a1) sm += v[m].x;
                        if (j + 1 >= a0 && j + 1 <= a1) sm += v[m].y;
                    }
                }
            } else {
                for (int i = s0 + lane; i <= s1; i += 256) {      // four loads in flight
                    const float q0 = x[i], q1 = i + 64 <= s1 ? x[i + 64] : 0.f, q2 = i + 128 <= s1 ? x[i + 128] : 0.f,
                                q3 = i + 192 <= s1 ? x[i + 192] : 0.f;
                    sm += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
                }
            }
            local_mean = group_sum<64>(sm) / (2.0 * P.nsamp_period);
        }
        // the windowed frame; the local peak over half a longest period around the window centre
        constexpr int WCH = R == 32 ? 8 : 4;
        double pk = 0.0;
#pragma unroll
        for (int mc = 0; mc < R; mc += WCH) {                     // WCH registers' window loads in flight
            double2_t w2[WCH];
#pragma unroll
            for (int u = 0; u < WCH; ++u) {
                const int j = 128 * (mc + u) + 2 * lane;
                w2[u] = *reinterpret_cast<const double2_t*>(win + (j < nw ? j : nw - 2));      // nw is even
            }
#pragma unroll
            for (int u = 0; u < WCH; ++u) {
                const int m = mc + u, jb = 128 * m, j = jb + 2 * lane;
                const bool on = j < nw;
                const double e0 = on ? (v[m].x - local_mean) * w2[u].x : 0.0, e1 = on ? (v[m].y - local_mean) * w2[u].y : 0.0;
                if (jb + 127 >= pa && jb < pb) {                  // uniform
                    if (j >= pa && j < pb) pk = fmax(pk, fabs(e0));
                    if (j + 1 >= pa && j + 1 < pb) pk = fmax(pk, fabs(e1));
                }
                v[m] = cplx{e0, e1};
            }
            asm volatile("" ::: "memory");
        }
        const double local_peak = wave_max_dpp(pk);
        const double intensity = gp > 0.0 ? (local_peak > gp ? 1.0 : local_peak / gp) : 0.0;
        if (P.debug_stop == 1) continue;
        // transform, |X|^2 repacked (conjugated), transform: r[2 k] = Re, r[2 k + 1] = -Im of element k
        wave_fft<R>(v, lds, lane, w_s, w_b);
        {
            ac_spec_store<R>(v, mem, lane);
            wave_sync();
            const cplx y_half = ac_spec_pairs<R>(v, mem, lane, ld_tw(tw, lane));
            wave_sync();
            ac_spec_load<R>(v, mem, lane, y_half);
            wave_sync();
        }
        // (the powers of the base twiddles are recomputed: kept from the first transform they would cost 120 registers)
        asm volatile("" : "+v"(lane));
        w_s = ld_tw(tw, 2 * lane);
        w_b = ld_tw(tw, (lane % L2) * (N / 64));
        wave_fft<R>(v, lds, lane, w_s, w_b);
        if (P.debug_stop == 2) continue;
        const double r0 = readlane_f64(v[0].x, 0);
        if (lane == 0) { rb[0] = 1.0; rb[L + 1] = intensity; }
        double wa[8], wb[8];                                      // L <= 1023: lags 2 (lane + 64 m), m < 8; loads first
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int l0 = 2 * (lane + 64 * m);
            wa[m] = wr[l0 < L ? l0 : L];
            wb[m] = wr[l0 + 1 < L ? l0 + 1 : L];
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int l0 = 2 * (lane + 64 * m);
            if (128 * m > L) break;                               // uniform
            if (l0 >= 1 && l0 <= L) rb[l0] = r0 > 0.0 ? v[m].x * fast_rcp(r0 * wa[m]) : 0.0;
            if (l0 + 1 <= L) rb[l0 + 1] = r0 > 0.0 ? -v[m].y * fast_rcp(r0 * wb[m]) : 0.0;
        }
    }
}

template <int R>
__global__ __launch_bounds__(64, R == 32 ? 2 : 3) void pitch_cc_wave_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                           const double* __restrict__ gpeak, const PitchParams P,
                                                           const double2_t* __restrict__ tw, double* __restrict__ rbuf,
                                                           int rstride, int max_frames) {
    using namespace wfft;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    double* s_sy = lds + Plan<R>::LDS_DOUBLES;                 // [L + 1] sumy2(l)
    const ClipInfo c = ci[blockIdx.y];
    const int f0 = blockIdx.x * WF_FRAMES;
    if (f0 >= c.n_frames) return;
    const int lane_ = threadIdx.x;
    constexpr int S = 64 * R, H = R / 2, L2 = 64 / R, L2H = 64 / H;   // tw = W_S^k, k < S / 2; transforms of S, then S / 2 points
    const int nw = P.nsamp_window, L = P.max_lag;
    const int seg_len = nw + L + 1;
    const float* x = wav + c.sample_off;
    const int n = c.n_samples;
    const double gp = gpeak[blockIdx.y];
    LdsMem mem{lds};
    const int f1 = f0 + WF_FRAMES < c.n_frames ? f0 + WF_FRAMES : c.n_frames;
    int pa = P.half_window - P.half_period, pb = P.half_window + P.half_period;
    pa = pa < 0 ? 0 : pa;
    pb = pb > nw ? nw : pb;
#pragma unroll 1
    for (int f = f0; f < f1; ++f) {
        int lane = lane_;                                          // redefined per frame: see pitch_ac_wave_kernel
        asm volatile("" : "+v"(lane));
        double* rb = rbuf + ((int64_t)blockIdx.y * max_frames + f) * rstride;   // r[0..L], then the intensity
        const double t = c.t1 + f * P.dt;
        const int left = (int)low_index(t, c.x1), right = left + 1;
        // Praat: startTime = t - 0.5 * (1 / minimumPitch + dt_window), dt_window = periods / minimumPitch
        const double start_time = t - 0.5 * (1.0 / P.min_pitch + P.dt_window);
        int64_t start64 = low_index(start_time, c.x1);
        if (start64 < 0) start64 = 0;
        const int start = (int)start64;
        int span = L + nw;
        if (span > n - start) span = n - start;
        const int loc_max_lag = span - nw;
        const int avail = n - start < seg_len ? n - start : seg_len;      // samples of the segment inside the sound
        // every load of the frame first: the raw segment b[j], j = lane + 64 m, and the samples nw behind the first L of them
        // (for the running sum below)
        // (unconditional loads on clamped indices, zeroed afterwards: see pitch_ac_wave_kernel)
        cplx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            int gi = start + lane + 64 * m;
            gi = gi > n - 1 ? n - 1 : gi;
            v[m] = cplx{0.0, (double)x[gi]};
        }
        float tail[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {                             // L <= 1023
            int gi = start + lane + 64 * q + nw;
            gi = gi > n - 1 ? n - 1 : gi;
            tail[q] = x[gi];
        }
        // local mean over one longest period to each side (divisor 2*nsamp_period as in Praat), from the registers when the
        // segment holds that range
        double local_mean;
        {
            int s0 = right - P.nsamp_period, s1 = left + P.nsamp_period;
            s0 = s0 < 0 ? 0 : (s0 > n - 1 ? n - 1 : s0);
            s1 = s1 < 0 ? 0 : (s1 > n - 1 ? n - 1 : s1);
            double sm = 0.0;
            if (s0 >= start && s1 < start + avail) {
                const int a0 = s0 - start, a1 = s1 - start;
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    const int jb = 64 * m, j = jb + lane;
                    if (jb + 63 >= a0 && jb <= a1 && j >= a0 && j <= a1) sm += v[m].y;
                }
            } else {
                for (int i = s0 + lane; i <= s1; i += 256) {      // four loads in flight
                    const float q0 = x[i], q1 = i + 64 <= s1 ? x[i + 64] : 0.f, q2 = i + 128 <= s1 ? x[i + 128] : 0.f,
                                q3 = i + 192 <= s1 ? x[i + 192] : 0.f;
                    sm += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
                }
            }
            local_mean = group_sum<64>(sm) / (2.0 * P.nsamp_period);
        }
        // z = a + i b: b = the segment minus the mean, a = its first nw samples; the local peak and sumx2 = sum a^2 on the way
        double pk = 0.0, sx = 0.0;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int jb = 64 * m, j = jb + lane;
            double e = 0.0, a = 0.0;
            if (j < avail) e = v[m].y - local_mean;
            if (j < nw) a = e;
            if (jb + 63 >= pa && jb < pb && j >= pa && j < pb) pk = fmax(pk, fabs(e));
            sx = fma(a, a, sx);
            v[m] = cplx{a, e};
        }
        const double local_peak = wave_max_dpp(pk);
        const double intensity = gp > 0.0 ? (local_peak > gp ? 1.0 : local_peak / gp) : 0.0;
        const double sumx2 = group_sum<64>(sx);
        // sumy2(l) = sum_{j=l}^{l+nw-1} b_j^2 = sumx2 + sum_{i<l} (b_{i+nw}^2 - b_i^2): a running sum over the L lags, 64 at a time
        {
            double carry = 0.0;
            if (lane == 0) s_sy[0] = sumx2;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (64 * q >= L) break;                            // uniform
                const int i = lane + 64 * q;
                double d = 0.0;
                if (i < L) {
                    const double u = i + nw < avail ? (double)tail[q] - local_mean : 0.0, w0 = v[q].y;
                    d = u * u - w0 * w0;
                }
                const double sc = wave_scan_incl(d, lane);
                if (i < L) s_sy[i + 1] = sumx2 + (carry + sc);
                carry += readlane_f64(sc, 63);
            }
        }
        if (P.debug_stop == 1) continue;
        const cplx w_s1 = ld_tw(tw, lane);                        // W_S^lane; W_64^(lane % L2) = W_S^(R (lane % L2))
        wave_fft<R>(v, lds, lane, w_s1, ld_tw(tw, (lane % L2) * R));
        // C = conj(A) B repacked into the S / 2 points of the transform back (wave_fft.h)
        cplx y[H];
        cc_spec_store<R>(v, mem, lane);
        wave_sync();
        const cplx y_half = cc_spec_pairs<R>(v, y, mem, lane, w_s1);
        wave_sync();
        cc_spec_load<R>(y, mem, lane, y_half);
        wave_sync();
        wave_fft<H>(y, lds, lane, ld_tw(tw, 2 * lane), ld_tw(tw, (lane % L2H) * R));   // W_(S/2)^lane, W_64^(lane % (2 L2))
        if (P.debug_stop == 2) continue;
        // r[2 k] = Re, r[2 k + 1] = -Im of element k, times S; normalised by sqrt(sumx2 sumy2(l))
        if (lane == 0) { rb[0] = 1.0; rb[L + 1] = intensity; }
        const double inv_n = 1.0 / (double)S;
        constexpr int MO = H < 8 ? H : 8;                         // L <= 1023: lags 2 (lane + 64 m), m < 8
        double ya[MO], yb[MO];
#pragma unroll
        for (int m = 0; m < MO; ++m) {
            const int l0 = 2 * (lane + 64 * m);
            ya[m] = s_sy[l0 < L ? l0 : L];
            yb[m] = s_sy[l0 + 1 < L ? l0 + 1 : L];
        }
#pragma unroll
        for (int m = 0; m < MO; ++m) {
            const int l0 = 2 * (lane + 64 * m);
            if (128 * m > L) break;                               // uniform
            if (l0 >= 1 && l0 <= L) {
                const double den = sumx2 * ya[m];
                rb[l0] = (l0 <= loc_max_lag && den > 0.0) ? (y[m].x * inv_n) * fast_rsqrt(den) : 0.0;
            }
            if (l0 + 1 <= L) {
                const double den = sumx2 * yb[m];
                rb[l0 + 1] = (l0 + 1 <= loc_max_lag && den > 0.0) ? (-y[m].y * inv_n) * fast_rsqrt(den) : 0.0;
            }
        }
        wave_sync();                                              // the next frame rewrites s_sy
    }
}

// Kernel 2 of 2: one wave per frame.  Reads the frame's normalised correlation row, finds the local maxima,
// estimates them (parabola + sinc 30), builds the candidate list(s) with Praat's replacement rule and refines
// every kept candidate with Brent's method.  All phases are single-wave, so nothing waits at a workgroup barrier
// and ~10 frames are resident per CU.
constexpr int CT = 64;          // threads of the candidate kernel
// Deferred refinement.  The Brent search of a frame keeps at most 15 of a wave's 64 lanes busy on a chain of dependent
// float64 operations, and the coefficient build of a cell whose depth the array ends clip needs that cell's own table.
// With `hdr` given the candidate kernel therefore stops at the candidate lists: it leaves a 128-byte record per frame
// (flags, list lengths, the lists' lags) and - unless the cells' coefficients are built per cell by
// pitch_cell_coef_kernel (`grouped`) - the Chebyshev coefficients of the candidates' cells in the workspace;
// pitch_brent_kernel then refines one candidate per lane (the candidates of sixteen frames packed into a wave).
constexpr int HDR_INTS = 32;                    // [0] flags, [1] length of list A, [2] of list B, [4..11] lags of A (16 x u16),
                                                // [12..19] lags of B, [20..23] for every slot of B: the slot of A with the same lag
constexpr int PC_DOUBLES = MAXC * 2 * NCH;      // per frame and list: [slot][cell][coefficient]
constexpr int HDR_A_DEFER = 1, HDR_B_DEFER = 2, HDR_B_COPY = 4;
struct DeferArgs {
    int* hdr;            // nullptr: everything in the candidate kernel (the form before round 4, kept as the A/B reference)
    double* pc_a;        // coefficients of the list that is refined first (the lower voicing threshold of a dual pass)
    double* pc_b;        // coefficients of the other list of a dual pass (only written when it holds a lag the first lacks)
    int grouped;         // 1: pitch_cell_coef_kernel builds pc_a
};
__global__ __launch_bounds__(64) void pitch_cand_kernel(const ClipInfo* __restrict__ ci, const double* __restrict__ gpeak,
                                                        const PitchParams P, const double* __restrict__ rbuf, int rstride,
                                                        int max_frames, FrameOut* __restrict__ out,
                                                        FrameOut* __restrict__ out2, const double* __restrict__ cheb,
                                                        const DeferArgs DA) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const ClipInfo c = ci[blockIdx.y];
    const int f = blockIdx.x;
    if (f >= c.n_frames) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = 0;
    const int L = P.is_cc ? P.max_lag : P.brent_ixmax;
    const int RC = P.brent_ixmax;                   // centre index of r
    const int RN = 2 * P.brent_ixmax + 1;
    // r is Praat's symmetric array of 2 ixmax + 1 lags, but only |lag| <= L is non-zero and nothing reads beyond
    // the candidates' lags widened by the interpolation depth (and never past 2 L + 1): only that range is stored,
    // through a base pointer shifted so that the indices stay Praat's
    // (with the per-cell coefficient kernel behind it, this kernel reads no further than its depth-30 first estimates)
    int r_lo, r_hi;
    pitch_r_range(RC, L, P.min_lag, P.max_lag, (DA.hdr && DA.grouped) ? 30 : P.refine_depth, &r_lo, &r_hi);
    double* r_store = reinterpret_cast<double*>(smem_raw);
    double* r = r_store - (RC + r_lo);
    double* s_mfreq = r_store + ((r_hi - r_lo + 2) & ~1);   // [MAX_MAXIMA]
    double* s_mstr = s_mfreq + MAX_MAXIMA;          // [MAX_MAXIMA]
    double* s_mloc = s_mstr + MAX_MAXIMA;           // [MAX_MAXIMA] strength - octave cost (Praat's "local strength")
    double* s_cf = s_mloc + MAX_MAXIMA;             // [MAXC]
    double* s_cs = s_cf + MAXC;                     // [MAXC]
    double* s_cloc = s_cs + MAXC;                   // [MAXC]
    int* s_maxlag = reinterpret_cast<int*>(s_cloc + MAXC);   // [MAX_MAXIMA]
    int* s_place = s_maxlag + MAX_MAXIMA;           // [MAXC]
    int* s_place2 = s_place + MAXC;                 // [MAXC]  second (lower-threshold) list
    int* s_cnt = s_place2 + MAXC;                   // [0] = nmax, [1] = ncand, [2] = ncand2, [3] = missing
    double* s_cf2 = reinterpret_cast<double*>(s_cnt + 4);   // [MAXC]
    double* s_cs2 = s_cf2 + MAXC;                   // [MAXC]
    double* s_cloc2 = s_cs2 + MAXC;                 // [MAXC]
    double* s_part = s_cloc2 + MAXC;                // [MAXC][2][NCH] Chebyshev coefficients of the candidates' cells
#define s_nmax s_cnt[0]
#define s_ncand s_cnt[1]

    FrameOut* o = out + c.frame_off + f;
#define RSAF_PITCH_DBG_STOP(k)                                                                       \
    if (P.debug_stop == (k)) {                                                                       \
        if (tid == 0) { o->intensity = 0.0; o->ncand = 1.0; }                                        \
        if (tid < MAXC) { o->freq[tid] = 0.0; o->strength[tid] = 0.0; }                              \
        return;                                                                                      \
    }
    RSAF_PITCH_DBG_STOP(1)
    RSAF_PITCH_DBG_STOP(2)
    const double* rb = rbuf + ((int64_t)blockIdx.y * max_frames + f) * rstride;
    const double intensity = rb[L + 1];
    const double gp = gpeak[blockIdx.y];
    // the row r[0..L] mirrored into Praat's symmetric array: ten loads in flight (unconditional, on clamped lags) before
    // the first store - one load per loop turn crossed the memory latency ten to fifteen times per frame
    for (int j0 = RC + r_lo + tid; j0 <= RC + r_hi; j0 += 10 * CT) {
        double v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int j = j0 + u * CT;
            const int l = j >= RC ? j - RC : RC - j;
            v[u] = rb[l <= L ? l : L];
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int j = j0 + u * CT;
            const int l = j >= RC ? j - RC : RC - j;
            if (j <= RC + r_hi) r[j] = l <= L ? v[u] : 0.0;
        }
    }
    if (tid == 0) s_nmax = 0;
    __syncthreads();
    // ---- local maxima in ascending lag order (wave 0, ballot + prefix) ----
    const bool dual = out2 != nullptr && P.voicing_thr2 >= 0.0;
    const double thr_low = dual && P.voicing_thr2 < P.voicing_thr ? P.voicing_thr2 : P.voicing_thr;
    const int lag_lo = P.min_lag > 2 ? P.min_lag : 2;
    int lag_hi = P.max_lag - 1;
    if (lag_hi > P.brent_ixmax - 1) lag_hi = P.brent_ixmax - 1;
    if (wv == 0) {
        int count = 0;
        for (int base = lag_lo; base <= lag_hi; base += 64) {
            const int l = base + lane;
            bool ok = false;
            if (l <= lag_hi) {
                const double v = r[RC + l];
                ok = (v > 0.5 * thr_low) && (v > r[RC + l - 1]) && (v >= r[RC + l + 1]);
            }
            const unsigned long long m = __ballot(ok);
            const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
            if (ok && pos < MAX_MAXIMA) s_maxlag[pos] = l;
            count += __popcll(m);
        }
        if (lane == 0) s_nmax = count < MAX_MAXIMA ? count : MAX_MAXIMA;
    }
    __syncthreads();
    RSAF_PITCH_DBG_STOP(3)
    const int nmax = s_nmax;
    const int nz_lo = RC - L, nz_hi = RC + L;
    // ---- first estimate of every maximum: parabolic position, sinc(30) strength (16 maxima per round) ----
    const int l16 = lane & 15, gidx = lane >> 4;
    for (int mb = 0; mb < nmax; mb += CT / 16) {
        const int m = mb + gidx;
        const bool live = m < nmax;
        const int l = s_maxlag[live ? m : 0];
        const double y0 = r[RC + l - 1], y1 = r[RC + l], y2 = r[RC + l + 1];
        const double dr = 0.5 * (y2 - y0), d2r = 2.0 * y1 - y0 - y2;
        const double fm = 1.0 / DXS / (l + dr / d2r);
        double st = sinc_group<16, false>(r, RN, RC + 1.0 / DXS / fm, 30, nz_lo, nz_hi, l16);
        if (st > 1.0) st = 1.0 / st;
        if (live && l16 == 0) { s_mfreq[m] = fm; s_mstr[m] = st; s_mloc[m] = st - P.octave_cost * log2(P.min_pitch / fm); }
    }
    __syncthreads();
    RSAF_PITCH_DBG_STOP(4)
    // ---- candidate list with replacement of the weakest (sequential in the maxima as in Praat, cooperative per step) ----
    // The maxima were collected for the lower of the two voicing thresholds; each list takes the maxima whose
    // correlation exceeds half its own threshold, in ascending lag order.  Lane z owns slot z of the list (its local
    // strength and the index of its maximum); lanes also hold the maxima's local strengths (lane l: maxima l and l + 64),
    // so the walk over the maxima reads them by v_readlane and the "weakest slot" is one wave reduction, redone only
    // after a replacement.  (The one-thread form re-read 14 slots from LDS per maximum: with the harmonicity pass'
    // threshold of 0 and up to 96 maxima that serial chain was a fifth of its frame time.)
    const double mloc0 = lane < nmax ? s_mloc[lane] : 0.0, mloc1 = lane + 64 < nmax ? s_mloc[lane + 64] : 0.0;
    const double mr0 = lane < nmax ? r[RC + s_maxlag[lane]] : -1.0, mr1 = lane + 64 < nmax ? r[RC + s_maxlag[lane + 64]] : -1.0;
    auto build_list = [&](double vthr, double* cf, double* cs, double* cloc, int* place_lag, int* ncand_out) {
        unsigned long long todo0 = __ballot(mr0 > 0.5 * vthr), todo1 = __ballot(mr1 > 0.5 * vthr);
        double my_loc = 0.0;
        int my_m = -1, nc = 1;
        double weakest = 2.0;
        int wplace = 0;
        bool known = false;                                   // weakest / wplace describe the current slots
        for (int half = 0; half < 2; ++half) {
            unsigned long long todo = half ? todo1 : todo0;
            while (todo) {
                const int bit = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int m = 64 * half + bit;
                const double loc_m = readlane_f64(half ? mloc1 : mloc0, bit);
                int place;
                if (nc < P.max_cand) {
                    place = nc++;
                } else {
                    if (!known) {                             // first minimum over slots 1 .. max_cand - 1 (Praat: strict <)
                        const double v = (lane >= 1 && lane < P.max_cand) ? my_loc : INFINITY;
                        double mn = v;
#pragma unroll
                        for (int o = 32; o >= 1; o >>= 1) mn = fmin(mn, __shfl_xor(mn, o, 64));
                        if (mn < 2.0) { weakest = mn; wplace = __ffsll((long long)__ballot(v == mn)) - 1; }
                        else { weakest = 2.0; wplace = 0; }
                        known = true;
                    }
                    place = loc_m <= weakest ? 0 : wplace;
                }
                if (place) {
                    if (lane == place) { my_loc = loc_m; my_m = m; }
                    known = false;
                }
            }
        }
        if (lane < MAXC) {
            const bool on = lane >= 1 && lane < nc && my_m >= 0;
            cf[lane] = on ? s_mfreq[my_m] : 0.0;
            cs[lane] = on ? s_mstr[my_m] : 0.0;
            cloc[lane] = on ? my_loc : 0.0;
            place_lag[lane] = on ? s_maxlag[my_m] : 0;
        }
        if (lane == 0) *ncand_out = nc;
    };
    build_list(P.voicing_thr, s_cf, s_cs, s_cloc, s_place, &s_cnt[1]);
    if (dual) build_list(P.voicing_thr2, s_cf2, s_cs2, s_cloc2, s_place2, &s_cnt[2]);
    __syncthreads();
    const int ncand = s_cnt[1];
    RSAF_PITCH_DBG_STOP(5)
    // ---- refine every kept candidate: maximise the sinc-interpolated correlation (16 at a time) ----
    // With every path cost zero (harmonicity pass) the path finder picks the strongest candidate of each
    // frame on its own, so a candidate far below the best first-pass strength can never be selected and
    // is left unrefined (the depth-30 and refined strengths differ by far less than the margin).
    // Lanes per candidate follow the candidate count (uniform per frame): few candidates (the usual AC case)
    // get a whole wave each, a full list gets 16 lanes each, so one or two rounds cover every frame.
    const int64_t wframe = (int64_t)blockIdx.y * max_frames + f;          // the frame's index in the workspace arrays
    // returns true when the list's refinement is left to pitch_brent_kernel
    auto refine_list = [&](int nc, const int* place_lag, double* cf, double* cs, double* pc_out) -> bool {
        if (DA.hdr && DA.grouped) return true;                           // per-cell tables: nothing to build here
        // Depth of the sinc interpolation on a cell whose left sample is b (0-based): Praat clips it to the samples that
        // exist on either side, min(depth, b + 1, n - b - 1), for BOTH halves of the kernel.  A frame whose cells all
        // have the full depth takes the shared table on the matrix pipe; a clipped cell (cc passes: lags within `depth` of
        // the end of the array) has its own table per depth (host-built, depths 3 .. depth - 1; 618 KB at depth 70) and
        // its 16 coefficients cost 2 d_c x 16 multiply-adds on the vector ALU - against ~15 Brent evaluations of the
        // 2 d_c-term sum with a reciprocal and a cosine per term in the direct form.  Depths below 3 (nearest / linear /
        // cubic in NUM_interpolate_sinc) and analyses without per-depth tables (depth 700: 140 tables of 41 KB would not
        // stay in L2) keep the direct form.
        bool use_cheb = cheb != nullptr;
        int n_clip_cols = 0;
        // lane k holds candidate k (k < 16: one DPP row): the scans over the candidates below are one LDS read per lane and
        // row reductions, not loops of dependent LDS reads
        const bool cand_lane = lane >= 1 && lane < nc;
        const int my_b0 = (cand_lane ? place_lag[lane] : 0) + RC - 1;
        auto row_min = [](int v) {
            v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));
            v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));
            v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
            v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false));
            return __builtin_amdgcn_readlane(v, 0);
        };
        if (use_cheb && !P.cheb_all_full) {
            int dc0 = P.refine_depth, dc1 = P.refine_depth;
            dc0 = dc0 < my_b0 + 1 ? dc0 : my_b0 + 1;
            dc0 = dc0 < RN - my_b0 - 1 ? dc0 : RN - my_b0 - 1;
            dc1 = dc1 < my_b0 + 2 ? dc1 : my_b0 + 2;
            dc1 = dc1 < RN - my_b0 - 2 ? dc1 : RN - my_b0 - 2;
            n_clip_cols = __popcll(__ballot(cand_lane && dc0 < P.refine_depth)) + __popcll(__ballot(cand_lane && dc1 < P.refine_depth));
            const int dmin = row_min(cand_lane ? min(dc0, dc1) : P.refine_depth);
            if (n_clip_cols > 0 && (!P.cheb_clipped || dmin < 3)) use_cheb = false;
        }
        if (use_cheb) {
            // Chebyshev coefficients of both cells of every candidate on the fp64 matrix pipe:
            //   P[j][(k, cell)] = sum_o cheb[o][j] * r[b_k + cell + o]      (16 coefficients x up to 30 columns x 2 d taps)
            // as v_mfma_f64_16x16x4: A[m = j][kk] = cheb[o + kk][j] (a lane's table load is the A operand as it is),
            // B[kk][n] = r[base_n + o + kk] with a per-lane base (column n = 2 (k - 1) + cell), one or two column tiles.
            // The vector-ALU form of this sum (8 FMAs and 8 LDS reads per table element and round of four candidates)
            // was 35-70 % of the harmonicity pass' frame time.
            double* s_P = s_part;                                    // [MAXC][2][NCH], the partial sums are dead by now
            const int d = P.refine_depth;
            const int kq = lane >> 4, nn = lane & 15;
            const int ncol = 2 * (nc - 1), tiles = (ncol + 15) >> 4;
            if (nc > 1 && ncol <= 12 && n_clip_cols == 0) {
                // Few candidates (the autocorrelation passes keep two or three): v_mfma_f64_4x4x4_4b instead - four independent
                // 4 x 4 x 4 blocks in 16 cycles where the 16 x 16 x 4 instruction takes 64 with 4-12 of its 16 columns in use.
                // Block = (lane % 16) / 4 holds coefficients 4 blk .. 4 blk + 3 (A[blk][i][k] in lane 16 k + 4 blk + i: the same table
                // element as the wide instruction's A operand), every block gets the same B (B[blk][k][j] in lane 16 k + 4 blk + j:
                // tap k of column j = lane % 4), D[blk][i][j] comes back in lane 16 i + 4 blk + j (tools/micro/mfma_f64_4x4_probe.hip).
                const int bmin = row_min(cand_lane ? my_b0 : 0x7fffffff), bmax = -row_min(cand_lane ? -my_b0 : 0x7fffffff);
                const int groups = (ncol + 3) >> 2;
                int rb4[3];
                bool cf4[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int n = 4 * g + (lane & 3), k = 1 + (n >> 1);
                    rb4[g] = place_lag[k < nc ? k : 1] + RC - 1 + (n & 1);
                    cf4[g] = n < ncol;                                   // (no clipped cell in this frame: they keep the wide path)
                }
                int o_lo = nz_lo - (bmax + 1), o_hi = nz_hi - bmin;
                o_lo = o_lo < -(d - 1) ? -(d - 1) : o_lo;
                o_hi = o_hi > d ? d : o_hi;
                double a4[3] = {0.0, 0.0, 0.0};
                const double* ctab = cheb + (int64_t)(d - 1) * NCH + nn;
                int o = o_lo;
                for (; o + 31 <= o_hi; o += 32) {
                    double cw[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) cw[u] = ctab[(int64_t)(o + 4 * u + kq) * NCH];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int oo = o + 4 * u + kq;
                        a4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(cw[u], cf4[0] ? r[rb4[0] + oo] : 0.0, a4[0], 0, 0, 0);
                        if (groups > 1) a4[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(cw[u], cf4[1] ? r[rb4[1] + oo] : 0.0, a4[1], 0, 0, 0);
                        if (groups > 2) a4[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(cw[u], cf4[2] ? r[rb4[2] + oo] : 0.0, a4[2], 0, 0, 0);
                    }
                }
                for (; o <= o_hi; o += 4) {
                    const int oo = o + kq, oc = oo <= o_hi ? oo : o_hi;          // taps past o_hi contribute zero
                    const double cw = oo <= o_hi ? ctab[(int64_t)oc * NCH] : 0.0;
                    a4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(cw, cf4[0] ? r[rb4[0] + oc] : 0.0, a4[0], 0, 0, 0);
                    if (groups > 1) a4[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(cw, cf4[1] ? r[rb4[1] + oc] : 0.0, a4[1], 0, 0, 0);
                    if (groups > 2) a4[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(cw, cf4[2] ? r[rb4[2] + oc] : 0.0, a4[2], 0, 0, 0);
                }
                const int coef = (nn & 12) + kq;                             // 4 blk + i
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int n = 4 * g + (lane & 3), k = 1 + (n >> 1);
                    if (g < groups && n < ncol) s_P[(k * 2 + (n & 1)) * NCH + coef] = a4[g];
                }
            } else if (nc > 1) {
                const int bmin = row_min(cand_lane ? my_b0 : 0x7fffffff), bmax = -row_min(cand_lane ? -my_b0 : 0x7fffffff);
                int rbase[2];
                bool colfull[2];                                      // clipped cells are rebuilt below: their B operand is zero here
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    const int n = 16 * T + nn, k = 1 + (n >> 1);
                    rbase[T] = place_lag[k < nc ? k : 1] + RC - 1 + (n & 1);   // 0-based left sample of the cell
                    colfull[T] = n_clip_cols == 0 || (rbase[T] + 1 >= d && RN - rbase[T] - 1 >= d);
                }
                // r is zero outside [nz_lo, nz_hi]: taps that reach no candidate's non-zero range are skipped
                int o_lo = nz_lo - (bmax + 1), o_hi = nz_hi - bmin;
                o_lo = o_lo < -(d - 1) ? -(d - 1) : o_lo;
                o_hi = o_hi > d ? d : o_hi;
                double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
                const double* ctab = cheb + (int64_t)(d - 1) * NCH + nn;
                // eight table loads (L2-resident, ~500 cycles each) are issued together before they are consumed
                int o = o_lo;
                for (; o + 31 <= o_hi; o += 32) {
                    double cw[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) cw[u] = ctab[(int64_t)(o + 4 * u + kq) * NCH];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int oo = o + 4 * u + kq;
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(cw[u], colfull[0] ? r[rbase[0] + oo] : 0.0, acc0, 0, 0, 0);
                        if (tiles > 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(cw[u], colfull[1] ? r[rbase[1] + oo] : 0.0, acc1, 0, 0, 0);
                    }
                }
                for (; o <= o_hi; o += 4) {
                    const int oo = o + kq, oc = oo <= o_hi ? oo : o_hi;          // taps past o_hi contribute zero
                    const double cw = oo <= o_hi ? ctab[(int64_t)oc * NCH] : 0.0;
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(cw, colfull[0] ? r[rbase[0] + oc] : 0.0, acc0, 0, 0, 0);
                    if (tiles > 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(cw, colfull[1] ? r[rbase[1] + oc] : 0.0, acc1, 0, 0, 0);
                }
                // D layout: lane (kq, nn), register v -> row j = kq + 4 v, column nn
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    const int n = 16 * T + nn, k = 1 + (n >> 1);
                    if (T < tiles && k < nc) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) s_P[(k * 2 + (n & 1)) * NCH + kq + 4 * v] = T == 0 ? acc0[v] : acc1[v];
                    }
                }
                if (n_clip_cols > 0) {
                    // clipped cells: coefficient j = sum over the 2 d_c taps of the depth's own table; lane = (tap phase, j)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const double* clip_tabs = cheb + (int64_t)2 * d * NCH;        // tables of depths 1 .. d - 1, depth e at 16 e (e - 1)
                    const int jj = lane & 15, ph = lane >> 4;
                    for (int col = 0; col < ncol; ++col) {
                        const int k = 1 + (col >> 1), b = place_lag[k] + RC - 1 + (col & 1);
                        int dc = d;
                        dc = dc < b + 1 ? dc : b + 1;
                        dc = dc < RN - b - 1 ? dc : RN - b - 1;
                        if (dc >= d) continue;                           // uniform
                        const double* tab = clip_tabs + (int64_t)NCH * dc * (dc - 1) + jj;
                        double acc = 0.0;
                        int o = -(dc - 1) + ph;
                        for (; o + 12 <= dc; o += 16) {                   // four table loads in flight
                            double tw4[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) tw4[u] = tab[(int64_t)(o + 4 * u + dc - 1) * NCH];
#pragma unroll
                            for (int u = 0; u < 4; ++u) acc = fma(tw4[u], r[b + o + 4 * u], acc);
                        }
                        for (; o <= dc; o += 4) acc = fma(tab[(int64_t)(o + dc - 1) * NCH], r[b + o], acc);
                        acc += __shfl_xor(acc, 16, 64);
                        acc += __shfl_xor(acc, 32, 64);
                        if (ph == 0) s_P[(k * 2 + (col & 1)) * NCH + jj] = acc;
                    }
                }
            }
            __syncthreads();
            if (P.debug_stop == 6) return true;
            if (DA.hdr) {                                            // the coefficients travel to pitch_brent_kernel
                for (int i = 2 * NCH + tid; i < nc * 2 * NCH; i += CT) pc_out[i] = s_P[i];
                __syncthreads();
                return true;
            }
            if (tid < 64) {                                          // nc <= 16: one lane per candidate
                const int k = 1 + tid;
                const bool live = k < nc;
                double xm, ym;
                improve_max_cheb(s_P + (live ? k : 1) * 2 * NCH, place_lag[live ? k : 1] + RC, live, xm, ym);
                if (ym > 1.0) ym = 1.0 / ym;
                if (live) { cf[k] = 1.0 / DXS / (xm - RC); cs[k] = ym; }
            }
            __syncthreads();
            return false;
        }
        const int nref = nc - 1;
        const int span = P.refine_depth < 2 * L ? P.refine_depth : 2 * L;    // longest half kernel
        RefineArgs A{r, RN, RC, P.refine_depth, nz_lo, nz_hi, nc, P.refine_margin, place_lag, cf, cs};
        if (nref <= 4) { if (span >= 6 * 64) refine_candidates<64, true>(A, tid, CT); else refine_candidates<64, false>(A, tid, CT); }
        else if (nref <= 8) { if (span >= 6 * 32) refine_candidates<32, true>(A, tid, CT); else refine_candidates<32, false>(A, tid, CT); }
        else { if (span >= 6 * 16) refine_candidates<16, true>(A, tid, CT); else refine_candidates<16, false>(A, tid, CT); }
        __syncthreads();
        return false;
    };
    int* hdr = DA.hdr ? DA.hdr + wframe * HDR_INTS : nullptr;
    int flags = 0;
    if (!dual) {
        if (refine_list(ncand, s_place, s_cf, s_cs, DA.pc_a ? DA.pc_a + wframe * PC_DOUBLES : nullptr)) flags = HDR_A_DEFER;
        if (hdr) {
            if (tid < MAXC) reinterpret_cast<unsigned short*>(hdr + 4)[tid] = (unsigned short)s_place[tid];
            if (tid == 0) { hdr[0] = gp > 0.0 ? flags : 0; hdr[1] = ncand; hdr[2] = 0; }
        }
    } else {
        // refine the lower-threshold list (normally a superset), copy the shared candidates by lag, and only
        // when a candidate of the primary list is missing from it (both lists overflowed) refine that list too
        const int ncand2 = s_cnt[2];
        const bool low_is_second = P.voicing_thr2 < P.voicing_thr;
        int* pl_a = low_is_second ? s_place2 : s_place;   double* cf_a = low_is_second ? s_cf2 : s_cf;   double* cs_a = low_is_second ? s_cs2 : s_cs;
        int* pl_b = low_is_second ? s_place : s_place2;   double* cf_b = low_is_second ? s_cf : s_cf2;   double* cs_b = low_is_second ? s_cs : s_cs2;
        const int nc_a = low_is_second ? ncand2 : ncand, nc_b = low_is_second ? ncand : ncand2;
        const bool def_a = refine_list(nc_a, pl_a, cf_a, cs_a, DA.pc_a ? DA.pc_a + wframe * PC_DOUBLES : nullptr);
        if (def_a) flags |= HDR_A_DEFER;
        if (tid == 0) s_cnt[3] = 0;
        __syncthreads();
        int hit = 0;
        {
            const int pa_reg = tid < MAXC ? pl_a[tid] : 0;             // lane z holds the lag of candidate z of the refined list
            const int mine = (tid >= 1 && tid < nc_b) ? pl_b[tid] : -1;
            for (int z = 1; z < nc_a; ++z) if (__builtin_amdgcn_readlane(pa_reg, z) == mine) hit = z;   // v_readlane: no LDS round trip per z
            if (tid >= 1 && tid < nc_b) {
                if (hit) { if (!def_a) { cf_b[tid] = cf_a[hit]; cs_b[tid] = cs_a[hit]; } } else atomicAdd(&s_cnt[3], 1);
            }
        }
        __syncthreads();
        if (s_cnt[3] > 0) {
            // restore the first estimates of the list (entries copied above hold refined values) and refine it whole
            if (tid >= 1 && tid < nc_b) {
                for (int m = 0; m < nmax; ++m) if (s_maxlag[m] == pl_b[tid]) { cf_b[tid] = s_mfreq[m]; cs_b[tid] = s_mstr[m]; }
            }
            __syncthreads();
            if (refine_list(nc_b, pl_b, cf_b, cs_b, DA.pc_b ? DA.pc_b + wframe * PC_DOUBLES : nullptr)) flags |= HDR_B_DEFER;
        } else if (def_a) {
            flags |= HDR_B_COPY;
        }
        if (hdr) {
            if (tid < MAXC) {
                reinterpret_cast<unsigned short*>(hdr + 4)[tid] = (unsigned short)pl_a[tid];
                reinterpret_cast<unsigned short*>(hdr + 12)[tid] = (unsigned short)pl_b[tid];
                reinterpret_cast<unsigned char*>(hdr + 20)[tid] = (unsigned char)hit;
            }
            if (tid == 0) { hdr[0] = gp > 0.0 ? flags : 0; hdr[1] = nc_a; hdr[2] = nc_b; }
        }
        if (tid == 0) {
            FrameOut* o2 = out2 + c.frame_off + f;
            o2->intensity = intensity;
            o2->ncand = gp > 0.0 ? (double)ncand2 : 1.0;
        }
        if (tid < MAXC) {
            FrameOut* o2 = out2 + c.frame_off + f;
            const bool on = tid < ncand2 && gp > 0.0;
            o2->freq[tid] = on ? s_cf2[tid] : 0.0;
            o2->strength[tid] = on ? s_cs2[tid] : 0.0;
        }
    }
    if (tid == 0) {
        o->intensity = intensity;
        o->ncand = gp > 0.0 ? (double)ncand : 1.0;
    }
    if (tid < MAXC) {
        const bool on = tid < ncand && gp > 0.0;
        o->freq[tid] = on ? s_cf[tid] : 0.0;
        o->strength[tid] = on ? s_cs[tid] : 0.0;
    }
}

// ---- deferred refinement, part 1: Chebyshev coefficients per CELL (analyses whose depth the array ends clip) ----------
// A cell is the interval between two samples of Praat's symmetric correlation array (index b = its left sample); the
// depth of the sinc interpolation on it is min(depth, b + 1, RN - b - 1) (NUM_interpolate_sinc), so in an analysis whose
// array is shorter than lag + depth every cell has its own depth and with it its own weight table - which the frame-wise
// kernel can only answer with the direct sum (a reciprocal and a cosine per tap and evaluation: 19.6 ms per 64 clips in the
// harmonicity pass at a 100 Hz floor, against 7-9 ms where one table serves every cell).  The table depends on b alone,
// only the taps that meet the non-zero lags |lag| <= L of r matter, and r[-lag] = r[lag] lets the table carry the sum of the
// two taps that meet a lag; in those terms
//     P_b[j][cell] = sum_m tab_b[m][j] * r_cell[m],        m = 0 .. L,
// i.e. one GEMM per b over all the cells of the batch that sit on b: a workgroup owns (b, a chunk of frames), keeps tab_b
// in LDS, scans the chunk's candidate lags for its two matches per frame (cell 0 of the candidate at lag b - RC + 1,
// cell 1 of the one at lag b - RC), and runs 16 cells at a time through v_mfma_f64_16x16x4 (A = the table from LDS,
// B = the cells' correlation rows from the workspace).  The host builds the tables (mshds.sinc_cell_tables).
// Workgroup id -> (b, chunk): all b of a chunk run on ONE XCD (ids are dealt round-robin to the 8 XCDs), so the chunk's rows
// - read 28 times over, by every cell of every candidate - stay in that XCD's L2.
constexpr int CELL_Q = 192;          // queue slots per wave (at most 15 left over + 128 new per scan step)
__global__ __launch_bounds__(256) void pitch_cell_coef_kernel(const ClipInfo* __restrict__ ci, int n_clips, int max_frames,
                                                              const int* __restrict__ hdr, const double* __restrict__ rbuf,
                                                              int rstride, int L, int RC, const double* __restrict__ tabs,
                                                              int b_lo, int n_b, int ntap_pad, int chunk_frames, int n_chunks,
                                                              double* __restrict__ pc) {
    using namespace wfft;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* tab = reinterpret_cast<double*>(smem_raw);                               // [ntap_pad][NCH]
    unsigned* qall = reinterpret_cast<unsigned*>(tab + (size_t)ntap_pad * NCH);      // [4][CELL_Q]
    const int64_t wid = blockIdx.x;
    const int64_t q = wid >> 3;
    const int bi = (int)(q % n_b);
    const int64_t chunk = (q / n_b) * 8 + (wid & 7);
    if (chunk >= n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned* qb = qall + wv * CELL_Q;
    const int b = b_lo + bi;
    const int lag0 = b - RC + 1, lag1 = b - RC;                // the lags whose cell 0 / cell 1 is this b
    const int64_t total = (int64_t)n_clips * max_frames;
    const int per_wave = chunk_frames / 4;
    const int64_t g_begin = chunk * chunk_frames + (int64_t)wv * per_wave;
    const int64_t g_end = g_begin + per_wave < total ? g_begin + per_wave : total;
    const int nn = lane & 15, kq = lane >> 4;
    // a lane's frame record: flags, list length, the 16 lags (zeros for a frame that does not exist or has nothing deferred)
    struct Rec { int nc; int lw[8]; };
    auto load_rec = [&](int64_t g) {
        Rec r;
        r.nc = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.lw[i] = 0;
        if (g < g_end) {
            const int clip = (int)(g / max_frames), fr = (int)(g - (int64_t)clip * max_frames);
            if (fr < ci[clip].n_frames) {
                const int* h = hdr + g * HDR_INTS;
                const int flags = h[0], nc = h[1];
#pragma unroll
                for (int i = 0; i < 8; ++i) r.lw[i] = h[4 + i];
                r.nc = (flags & HDR_A_DEFER) ? nc : 0;
            }
        }
        return r;
    };
    Rec rec = load_rec(g_begin + lane);                        // in flight while the table arrives
    {
        // the cell's table -> LDS, sixteen bytes per lane and load, eight loads in flight
        const double2_t* src = reinterpret_cast<const double2_t*>(tabs + (int64_t)bi * ntap_pad * NCH);
        double2_t* dst = reinterpret_cast<double2_t*>(tab);
        const int n2 = ntap_pad * NCH / 2;
        for (int i0 = tid; i0 < n2; i0 += 8 * 256) {
            double2_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i0 + 256 * u < n2 ? i0 + 256 * u : n2 - 1];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + 256 * u < n2) dst[i0 + 256 * u] = v[u];
        }
    }
    __syncthreads();
    int count = 0;
    // sixteen cells through the matrix pipe: eight tap groups per batch, the next batch's operands (rows from L2, table from
    // LDS) are requested before the current batch's eight dependent MFMAs are issued
    auto run_tile = [&](int off, int n) {                      // queue entries off .. off + n - 1 (n <= 16)
        const unsigned e = qb[off + (nn < n ? nn : 0)];
        const int64_t g = e >> 5;
        const int k = (e >> 1) & 15, c = e & 1;
        const double* rb = rbuf + g * rstride;
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
        double av[2][8], bv[2][8];
        auto fetch = [&](int t0, double (&a8)[8], double (&b8)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 4 * u + kq;                 // the lag
                const int tc = t < ntap_pad ? t : ntap_pad - 1;
                b8[u] = rb[tc <= L ? tc : L];                  // lags beyond L only meet the zero rows that pad the table
                a8[u] = t < ntap_pad ? tab[tc * NCH + nn] : 0.0;
            }
        };
        fetch(0, av[0], bv[0]);
        for (int t0 = 0; t0 < ntap_pad; t0 += 64) {
            if (t0 + 32 < ntap_pad) fetch(t0 + 32, av[1], bv[1]);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][u], bv[0][u], acc, 0, 0, 0);
            if (t0 + 32 >= ntap_pad) break;
            if (t0 + 64 < ntap_pad) fetch(t0 + 64, av[0], bv[0]);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1][u], bv[1][u], acc, 0, 0, 0);
        }
        if (nn < n) {                                          // D: lane (kq, nn), register v -> coefficient kq + 4 v of cell nn
            double* dst = pc + (g * MAXC + k) * (2 * NCH) + c * NCH;
#pragma unroll
            for (int v = 0; v < 4; ++v) dst[kq + 4 * v] = acc[v];
        }
    };
    for (int64_t g0 = g_begin; g0 < g_end; g0 += 64) {
        const int64_t g = g0 + lane;
        int m0 = 0, m1 = 0;
#pragma unroll
        for (int k = 1; k < MAXC; ++k) {                       // a list holds a lag once: at most one match per cell side
            const int lg = (rec.lw[k >> 1] >> (16 * (k & 1))) & 0xffff;
            if (k < rec.nc && lg == lag0) m0 = k;
            if (k < rec.nc && lg == lag1) m1 = k;
        }
        rec = load_rec(g + 64);                                // the next step's records travel while this step's tiles run
        const unsigned long long b0 = __ballot(m0 != 0), b1 = __ballot(m1 != 0);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (m0) qb[count + __popcll(b0 & below)] = ((unsigned)g << 5) | ((unsigned)m0 << 1);
        count += __popcll(b0);
        if (m1) qb[count + __popcll(b1 & below)] = ((unsigned)g << 5) | ((unsigned)m1 << 1) | 1u;
        count += __popcll(b1);
        wave_sync();
        int done = 0;
        for (; count - done >= 16; done += 16) run_tile(done, 16);
        const int left = count - done;                         // < 16 entries move to the front
        const unsigned rest = lane < left ? qb[done + lane] : 0u;
        wave_sync();
        if (lane < left) qb[lane] = rest;
        count = left;
        wave_sync();
    }
    if (count > 0) run_tile(0, count);
}

// ---- deferred refinement, part 2: Brent's search, one candidate per lane ----------------------------------------------
// A wave takes BR_FRAMES consecutive frames of a clip and packs their deferred candidates (list A, then - where the second
// list of a dual pass has to be refined on its own - list B) into its lanes: an autocorrelation pass keeps two or three
// candidates per frame, so a frame-per-16-lanes mapping would leave four lanes in five idle through every iteration.
// A candidate of list A also writes the slots of list B that hold the same lag (HDR_B_COPY: the second threshold's list).
constexpr int BR_FRAMES = 16;
__global__ __launch_bounds__(64) void pitch_brent_kernel(const ClipInfo* __restrict__ ci, const int* __restrict__ hdr,
                                                         const double* __restrict__ pc_a, const double* __restrict__ pc_b,
                                                         int max_frames, int RC, FrameOut* __restrict__ out_a,
                                                         FrameOut* __restrict__ out_b) {
    const ClipInfo c = ci[blockIdx.y];
    const int f0 = blockIdx.x * BR_FRAMES;
    if (f0 >= c.n_frames) return;
    const int lane = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.y * max_frames + f0;
    int cnt_a = 0, cnt_b = 0;
    if (lane < BR_FRAMES && f0 + lane < c.n_frames) {
        const int* h = hdr + (g0 + lane) * HDR_INTS;
        const int flags = h[0];
        if (flags & HDR_A_DEFER) cnt_a = h[1] > 1 ? h[1] - 1 : 0;
        if (flags & HDR_B_DEFER) cnt_b = h[2] > 1 ? h[2] - 1 : 0;
    }
    const int tot = cnt_a + cnt_b;
    int incl = tot;
#pragma unroll
    for (int o = 1; o < BR_FRAMES; o <<= 1) { const int up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
    const int excl = incl - tot;
    const int T = __builtin_amdgcn_readlane(incl, BR_FRAMES - 1);
    for (int q0 = 0; q0 < T; q0 += 64) {
        const int q = q0 + lane;
        const bool live = q < T;
        int j = 0;                                            // the last frame whose first candidate is at or before q
#pragma unroll
        for (int jj = 1; jj < BR_FRAMES; ++jj) if (__builtin_amdgcn_readlane(excl, jj) <= q) j = jj;
        if (!live) j = 0;
        const int r = q - __shfl(excl, j, 64), na = __shfl(cnt_a, j, 64);
        const bool is_b = live && r >= na;
        const int slot = live ? (is_b ? r - na : r) + 1 : 1;
        const int64_t g = g0 + j;
        const int* h = hdr + g * HDR_INTS;
        const int lag = reinterpret_cast<const unsigned short*>(h + (is_b ? 12 : 4))[slot];
        const double* Pc = (is_b ? pc_b : pc_a) + (g * MAXC + slot) * (2 * NCH);
        double xm, ym;
        improve_max_cheb(Pc, lag + RC, live, xm, ym);
        if (ym > 1.0) ym = 1.0 / ym;
        const double fq = 1.0 / DXS / (xm - RC);
        if (live) {
            FrameOut* o = (is_b ? out_b : out_a) + c.frame_off + f0 + j;
            o->freq[slot] = fq;
            o->strength[slot] = ym;
            if (!is_b && out_b != nullptr && (h[0] & HDR_B_COPY)) {
                const int nc_b = h[2];
                FrameOut* o2 = out_b + c.frame_off + f0 + j;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned hw = (unsigned)h[20 + w];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = 4 * w + u;
                        if (t >= 1 && t < nc_b && (int)((hw >> (8 * u)) & 0xff) == slot) { o2->freq[t] = fq; o2->strength[t] = ym; }
                    }
                }
            }
        }
    }
}

// ---- Viterbi path finder: one wave per clip (lane = candidate of the current frame) -------------------
constexpr int PATH_CH = 32;          // frames whose own costs are computed at once (lane-parallel) ahead of the dependent steps
constexpr int PATH_RING = 4 * PATH_CH;   // frames of the LDS ring: the chunk being written lies >= 2 chunks behind the frames being read

__global__ __launch_bounds__(64) void path_kernel(const FrameOut* __restrict__ fr, const ClipInfo* __restrict__ ci,
                                                  double dt, double silence_thr, double voicing_thr, double octave_cost,
                                                  double octave_jump_cost, double vuv_cost, double ceiling,
                                                  unsigned char* __restrict__ psi, int* __restrict__ end_state) {
    // ring of PATH_RING frames: a frame's own costs at [(f mod PATH_RING) * MAXC + candidate]
    __shared__ double s_delta[PATH_RING * MAXC], s_logf[PATH_RING * MAXC];
    __shared__ unsigned char s_flag[PATH_RING * MAXC];       // bit 0: voiceless, bit 1: valid
    __shared__ double s_cur[2][MAXC];                        // path costs of the previous frame (double-buffered by frame parity)
    __shared__ __attribute__((aligned(16))) unsigned char s_psi[PATH_CH * MAXC];   // back pointers of the current chunk
    const ClipInfo c = ci[blockIdx.x];
    const int lane = threadIdx.x;
    const int nF = c.n_frames;
    if (nF <= 0) return;
    const FrameOut* F = fr + c.frame_off;
    unsigned char* P = psi + c.frame_off * MAXC;
    const double corr = 0.01 / dt;
    const double ojc = octave_jump_cost * corr, vuc = vuv_cost * corr;
    // The loop-carried chain is the vector of path costs alone.  (1) A frame's own costs (two log2 per candidate) do not
    // depend on the path: they are computed for PATH_CH frames at a time with every lane busy, one chunk ahead, into an LDS
    // ring; the global loads of the chunk after that are in flight while the dependent steps run.  (2) A step is the
    // 16 x 16 table of transitions: lane (j, q) = (lane >> 2, lane & 3) evaluates predecessors q, q + 4, q + 8, q + 12 of
    // candidate j (four independent evaluations), the quad combines them in two exchange steps (first maximum: ties go
    // to the lower predecessor, as Praat's strict > in ascending order does), lane (j, 0) publishes the new cost through
    // LDS.  One LDS round trip per frame instead of up to 15 dependent v_readlane rounds.
    auto clampn = [](double nc) { const int v = (int)nc; return v < 0 ? 0 : (v > MAXC ? MAXC : v); };
    constexpr int PPLN = PATH_CH * MAXC / 64;                  // (frame, candidate) pairs per lane
    constexpr int RING = PATH_RING;
    double lfq[PPLN], lst[PPLN], lnc[PPLN], lin[PPLN];
    auto fetch = [&](int f0) {                                // frames [f0, f0 + PATH_CH) -> registers
#pragma unroll
        for (int u = 0; u < PPLN; ++u) {
            const int pr = lane + 64 * u, fi = f0 + (pr >> 4), cd = pr & (MAXC - 1);
            const int f = fi < nF ? fi : nF - 1;
            lfq[u] = F[f].freq[cd]; lst[u] = F[f].strength[cd]; lnc[u] = F[f].ncand; lin[u] = F[f].intensity;
        }
    };
    auto derive_to = [&](int f0) {                            // registers -> ring slots of frames [f0, f0 + PATH_CH)
        const int base = (f0 % RING) * MAXC;
#pragma unroll
        for (int u = 0; u < PPLN; ++u) {
            const int pr = lane + 64 * u, cd = pr & (MAXC - 1);
            const bool valid = cd < clampn(lnc[u]);
            const bool vl = !(lfq[u] > 0.0 && lfq[u] < ceiling);
            double unv = silence_thr <= 0.0 ? 0.0 : 2.0 - lin[u] / (silence_thr / (1.0 + voicing_thr));
            unv = voicing_thr + fmax(0.0, unv);
            s_delta[base + pr] = valid ? (vl ? unv : lst[u] - octave_cost * log2(ceiling / lfq[u])) : -1e300;
            s_logf[base + pr] = vl ? 0.0 : log2(lfq[u]);
            s_flag[base + pr] = (unsigned char)((vl ? 1 : 0) | (valid ? 2 : 0));
        }
    };
    // one wave per workgroup: wavefront-scope fences order the LDS traffic without draining the global stores (a
    // workgroup-scope release waits for vmcnt(0): with a global store per step that was most of the step)
    auto lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const int j = lane >> 2, q = lane & 3;
    fetch(0);
    derive_to(0);
    fetch(PATH_CH);
    lds_sync();
    if (q == 0) { s_cur[0][j] = s_delta[j]; s_psi[j] = 0; }   // frame 0: the path cost is the frame's own cost
    double cur = s_delta[j];
    for (int f0 = 0; f0 < nF; f0 += PATH_CH) {
        if (f0 + PATH_CH < nF) {                               // the next chunk's costs, then the loads of the one after it
            derive_to(f0 + PATH_CH);
            fetch(f0 + 2 * PATH_CH);
        }
        lds_sync();
        const int fend = f0 + PATH_CH < nF ? f0 + PATH_CH : nF;
#pragma unroll 1
        for (int f = f0 > 0 ? f0 : 1; f < fend; ++f) {
            const int me = (f % RING) * MAXC + j, pb = ((f - 1) % RING) * MAXC;
            const double delta = s_delta[me], logf = s_logf[me];
            const int fl = s_flag[me], vl = fl & 1, valid = (fl >> 1) & 1;
            const double* pcur = s_cur[(f - 1) & 1];
            double best = -INFINITY;
            int place = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c1 = q + 4 * u;
                const double pc = pcur[c1], pl = s_logf[pb + c1];
                const int pf = s_flag[pb + c1], pv = pf & 1, pval = (pf >> 1) & 1;
                const double tc = (pv && vl) ? 0.0 : ((pv || vl) ? vuc : ojc * fabs(pl - logf));
                const double v = pval ? pc - tc + delta : -INFINITY;
                if (v > best) { best = v; place = c1; }
            }
            {                                                  // the quad's maximum (two DPP quad permutes: VALU latency, no
                double ov = dpp_f64<0xB1>(best);               // LDS crossbar); ties: the lower predecessor
                int op = __builtin_amdgcn_update_dpp(0, place, 0xB1, 0xf, 0xf, false);
                if (ov > best || (ov == best && op < place)) { best = ov; place = op; }
                ov = dpp_f64<0x4E>(best);
                op = __builtin_amdgcn_update_dpp(0, place, 0x4E, 0xf, 0xf, false);
                if (ov > best || (ov == best && op < place)) { best = ov; place = op; }
            }
            if (best == -INFINITY) place = 0;                  // no valid predecessor: Praat leaves place at its initial 0
            if (!valid) best = -1e300;
            cur = best;
            if (q == 0) { s_cur[f & 1][j] = best; s_psi[(f - f0) * MAXC + j] = (unsigned char)place; }
            lds_sync();
        }
        // the chunk's back pointers leave as 16-byte rows
        if (lane < fend - f0)
            reinterpret_cast<uint4*>(P + (int64_t)f0 * MAXC)[lane] = reinterpret_cast<const uint4*>(s_psi)[lane];
        lds_sync();
    }
    // best end state: first maximum
    double bv = q == 0 ? cur : -INFINITY;
    int bi = j;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) end_state[blockIdx.x] = bi;
}

// backtrack in its own launch: the kernel boundary makes the psi stores of path_kernel visible.
// The back-pointer walk is a dependent chain of byte loads (state of frame f-1 = psi[f][state of frame f]).  The table
// is staged in LDS chunk by chunk (from the last frame backwards); inside a chunk the chain is cut into 256 segments:
// (1) every thread walks its segment for all 16 possible entry states at once (16 independent chains: the segment's
// map entry -> exit), (2) one thread threads the true state through the 256 maps, (3) every thread walks its segment
// again from its true entry state and records the states, (4) all threads gather the selected values.
// 2 x 12 + 256 dependent LDS reads per 3 072 frames instead of 3 072.
constexpr int BT_CHUNK = 3072;
__global__ __launch_bounds__(256) void backtrack_kernel(const FrameOut* __restrict__ fr, const ClipInfo* __restrict__ ci,
                                                        const unsigned char* __restrict__ psi,
                                                        const int* __restrict__ end_state, double* __restrict__ sel_freq,
                                                        double* __restrict__ sel_strength) {
    __shared__ __attribute__((aligned(16))) unsigned char s_psi[BT_CHUNK * MAXC];
    __shared__ unsigned char s_state[BT_CHUNK];
    __shared__ unsigned char s_map[256 * MAXC];
    __shared__ unsigned char s_entry[256];
    __shared__ int s_carry;
    const ClipInfo c = ci[blockIdx.x];
    if (c.n_frames <= 0) return;
    const int tid = threadIdx.x;
    const FrameOut* F = fr + c.frame_off;
    const unsigned char* P = psi + c.frame_off * MAXC;
    if (tid == 0) s_carry = end_state[blockIdx.x];
    for (int hi = c.n_frames; hi > 0; hi -= BT_CHUNK) {        // frames [lo, hi)
        const int lo = hi > BT_CHUNK ? hi - BT_CHUNK : 0, cnt = hi - lo;
        const uint4* src = reinterpret_cast<const uint4*>(P + (int64_t)lo * MAXC);   // MAXC == 16 bytes per frame
        uint4* dst = reinterpret_cast<uint4*>(s_psi);
        for (int i = tid; i < cnt; i += 256) dst[i] = src[i];
        __syncthreads();
        const int SL = (cnt + 255) / 256, nseg = (cnt + SL - 1) / SL;
        const int fb = tid * SL, ft = min(fb + SL, cnt) - 1;   // this thread's frames [fb, ft] of the chunk (top = ft)
        if (tid < nseg) {
            unsigned char x[MAXC];
#pragma unroll
            for (int e = 0; e < MAXC; ++e) x[e] = (unsigned char)e;
            for (int f = ft; f >= fb; --f) {
                if (lo + f > 0) {
#pragma unroll
                    for (int e = 0; e < MAXC; ++e) x[e] = s_psi[f * MAXC + x[e]];
                }
            }
#pragma unroll
            for (int e = 0; e < MAXC; ++e) s_map[tid * MAXC + e] = x[e];
        }
        __syncthreads();
        if (tid == 0) {
            int st = s_carry;                                   // state of frame hi-1
            for (int t = nseg - 1; t >= 0; --t) { s_entry[t] = (unsigned char)st; st = s_map[t * MAXC + st]; }
            s_carry = st;                                       // state of frame lo-1
        }
        __syncthreads();
        if (tid < nseg) {
            int st = s_entry[tid];
            for (int f = ft; f >= fb; --f) {
                s_state[f] = (unsigned char)st;
                if (lo + f > 0) st = s_psi[f * MAXC + st];
            }
        }
        __syncthreads();
        for (int i = tid; i < cnt; i += 256) {
            const int st = s_state[i];
            sel_freq[c.frame_off + lo + i] = F[lo + i].freq[st];
            sel_strength[c.frame_off + lo + i] = F[lo + i].strength[st];
        }
        __syncthreads();
    }
}

// ---- per-clip statistics of a selected pitch track ------------------------------------------------------
// out[clip][8]: 0 n_nonzero, 1 mean(nonzero), 2 std(nonzero, population), 3 mean after |z|<=2 filter,
//               4 n_voiced (0<f<ceiling), 5 mean Hz, 6 sd semitones (n-1), 7 count after filter
__global__ __launch_bounds__(64) void pitch_stats_kernel(const double* __restrict__ sel_freq, const ClipInfo* __restrict__ ci,
                                                         double ceiling, double* __restrict__ out) {
    const ClipInfo c = ci[blockIdx.x];
    const int lane = threadIdx.x;
    const double* f = sel_freq + c.frame_off;
    double n0 = 0, s0 = 0, nv = 0, sv = 0, sst = 0;
    for (int i = lane; i < c.n_frames; i += 64) {
        const double v = f[i];
        if (v != 0.0) { n0 += 1; s0 += v; }
        if (v > 0.0 && v < ceiling) { nv += 1; sv += v; sst += 12.0 * log2(v / 100.0); }
    }
    n0 = wave_sum_f64(n0); s0 = wave_sum_f64(s0); nv = wave_sum_f64(nv); sv = wave_sum_f64(sv); sst = wave_sum_f64(sst);
    const double m0 = n0 > 0 ? s0 / n0 : 0.0, mst = nv > 0 ? sst / nv : 0.0;
    double q0 = 0, qst = 0;
    for (int i = lane; i < c.n_frames; i += 64) {
        const double v = f[i];
        if (v != 0.0) q0 += (v - m0) * (v - m0);
        if (v > 0.0 && v < ceiling) { const double d = 12.0 * log2(v / 100.0) - mst; qst += d * d; }
    }
    q0 = wave_sum_f64(q0); qst = wave_sum_f64(qst);
    const double sd0 = n0 > 0 ? sqrt(q0 / n0) : 0.0;
    double nf = 0, sf = 0;
    for (int i = lane; i < c.n_frames; i += 64) {
        const double v = f[i];
        if (v != 0.0 && fabs((v - m0) / sd0) <= 2.0) { nf += 1; sf += v; }
    }
    nf = wave_sum_f64(nf); sf = wave_sum_f64(sf);
    if (lane == 0) {
        double* o = out + (int64_t)blockIdx.x * 8;
        const double qn = __longlong_as_double(0x7ff8000000000000LL);
        o[0] = n0; o[1] = m0; o[2] = sd0; o[3] = nf > 0 ? sf / nf : qn;
        o[4] = nv; o[5] = nv > 0 ? sv / nv : qn; o[6] = nv > 1 ? sqrt(qst / (nv - 1)) : qn; o[7] = nf;
    }
}

// ---- intensity statistics: energy mean, parabolic max / min -----------------------------------------------
__global__ __launch_bounds__(64) void intensity_stats_kernel(const double* __restrict__ db, const ClipInfo* __restrict__ ci,
                                                             double* __restrict__ out) {
    const ClipInfo c = ci[blockIdx.x];
    const int lane = threadIdx.x, n = c.n_frames;
    const double* y = db + c.frame_off;
    const double qn = __longlong_as_double(0x7ff8000000000000LL);
    if (n <= 0) {
        if (lane == 0) { out[blockIdx.x * 2] = qn; out[blockIdx.x * 2 + 1] = qn; }
        return;
    }
    double se = 0.0, mx = -INFINITY, mn = -INFINITY;      // mn holds the maximum of -y
    for (int i = lane; i < n; i += 64) {
        const double v = y[i];
        se += pow(10.0, v / 10.0);
        if (i == 0 || i == n - 1) { mx = fmax(mx, v); mn = fmax(mn, -v); }
        if (i > 0 && i < n - 1) {
            const double a = y[i - 1], b = y[i + 1];
            if (v > a && v >= b) {
                const double dy = 0.5 * (b - a), d2 = 2.0 * v - a - b;
                mx = fmax(mx, d2 != 0.0 ? v + 0.5 * dy * dy / d2 : v);
            }
            if (-v > -a && -v >= -b) {
                const double dy = 0.5 * (a - b), d2 = -2.0 * v + a + b;
                mn = fmax(mn, d2 != 0.0 ? -v + 0.5 * dy * dy / d2 : -v);
            }
        }
    }
    se = wave_sum_f64(se); mx = wave_max_f64(mx); mn = wave_max_f64(mn);
    if (lane == 0) {
        const double mean_db = 10.0 * log10(se / n);
        const double minv = -mn;
        out[blockIdx.x * 2] = mean_db;
        out[blockIdx.x * 2 + 1] = minv != 0.0 ? mx / minv : qn;
    }
}

// ---- HNR mean: 10 log10(r / (1 - r)) over voiced frames of a cc pitch track -------------------------------
__global__ __launch_bounds__(64) void hnr_stats_kernel(const double* __restrict__ sel_freq, const double* __restrict__ sel_str,
                                                       const ClipInfo* __restrict__ ci, double* __restrict__ out) {
    const ClipInfo c = ci[blockIdx.x];
    const int lane = threadIdx.x;
    double n = 0, s = 0;
    for (int i = lane; i < c.n_frames; i += 64) {
        const double f = sel_freq[c.frame_off + i], r = sel_str[c.frame_off + i];
        if (f != 0.0) {
            n += 1;
            s += r <= 1e-15 ? -150.0 : (r > 1.0 - 1e-15 ? 150.0 : 10.0 * log10(r / (1.0 - r)));
        }
    }
    n = wave_sum_f64(n); s = wave_sum_f64(s);
    if (lane == 0) out[blockIdx.x] = n > 0 ? s / n : __longlong_as_double(0x7ff8000000000000LL);
}

// ---- Gaussian-window spectrogram slice + spectral moments, gated by pitch definedness ----------------------
// one workgroup per frame; fp64 radix-2 FFT (in LDS, half length: real input) of the zero-padded windowed frame, bins 0..nbins-1 (bin width 1/(dx*nfft))
__global__ __launch_bounds__(512) void spec_moments_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                           const ClipInfo* __restrict__ pitch_ci, const double* __restrict__ sel_freq,
                                                           double pitch_dt, double ceiling, const double* __restrict__ win,
                                                           const double2* __restrict__ tw, int nsamp, int half, int nfft,
                                                           int nbins, double tstep, double fstep,
                                                           double* __restrict__ mom /* [frames][5]: ok, cog, sd, skew, kurt */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double2* a = reinterpret_cast<double2*>(smem_raw);          // nfft complex values of the FFT
    double* pw = reinterpret_cast<double*>(a + nfft / 2);        // nbins (the FFT works on nfft / 2 complex values)
    __shared__ double s_red[8][4];
    const ClipInfo c = ci[blockIdx.y];
    const int f = blockIdx.x;
    if (f >= c.n_frames) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwv = blockDim.x >> 6;
    double* o = mom + (c.frame_off + f) * 5;
    const double t = c.t1 + f * tstep;
    // gate: Pitch "Get value at time" defined iff the nearest pitch frame is voiced
    {
        const ClipInfo pc = pitch_ci[blockIdx.y];
        const double ireal = (t - pc.t1) / pitch_dt;
        const double il = floor(ireal);
        const int64_t near = (ireal - il < 0.5) ? (int64_t)il : (int64_t)il + 1;
        bool ok = near >= 0 && near < pc.n_frames;
        if (ok) {
            const double pf = sel_freq[pc.frame_off + near];
            ok = pf > 0.0 && pf < ceiling;
        }
        if (!ok) {
            if (tid == 0) o[0] = 0.0;
            return;
        }
    }
    const float* x = wav + c.sample_off;
    const int64_t start = low_index(t, c.x1) + 1 - half;
    // windowed frame, zero-padded to nfft.  Real input: one complex FFT of half the length over z[j] = x[2j] + i x[2j+1]
    // (stored bit-reversed for the in-place radix-2 passes), then X[k] = E + W^k O with E / O the even / odd parts.
    const int nthr = blockDim.x;
    int log2n = 0;
    while ((1 << log2n) < nfft) ++log2n;
    const int m = nfft >> 1, log2m = log2n - 1;
    for (int j = tid; j < m; j += nthr) {
        double v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int jj = 2 * j + e;
            v[e] = 0.0;
            if (jj < nsamp) {
                int64_t i = start + jj;
                i = i < 0 ? 0 : (i > c.n_samples - 1 ? c.n_samples - 1 : i);
                v[e] = (double)x[i] * win[jj];
            }
        }
        a[log2m ? (int)(__brev((unsigned)j) >> (32 - log2m)) : 0] = make_double2(v[0], v[1]);
    }
    __syncthreads();
    for (int st = 1; st <= log2m; ++st) {
        const int half_ = 1 << (st - 1), tstep_ = nfft >> st;
        for (int b = tid; b < (m >> 1); b += nthr) {
            const int grp = b >> (st - 1), p = b & (half_ - 1);
            const int i0 = (grp << st) + p, i1 = i0 + half_;
            const double2 w = tw[p * tstep_];
            const double2 u = a[i0], v = a[i1];
            const double tr = v.x * w.x - v.y * w.y, ti = v.x * w.y + v.y * w.x;
            a[i0] = make_double2(u.x + tr, u.y + ti);
            a[i1] = make_double2(u.x - tr, u.y - ti);
        }
        __syncthreads();
    }
    for (int k = tid; k < nbins; k += nthr) {              // nbins <= nfft / 2 + 1
        const double2 zk = a[k == m ? 0 : k], zc = a[k == 0 ? 0 : m - k];
        const double er = 0.5 * (zk.x + zc.x), ei = 0.5 * (zk.y - zc.y);
        const double orr = 0.5 * (zk.y + zc.y), oi = -0.5 * (zk.x - zc.x);
        const double2 w = k == m ? make_double2(-1.0, 0.0) : tw[k];
        const double re = er + w.x * orr - w.y * oi, im = ei + w.x * oi + w.y * orr;
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    double s0 = 0, s1 = 0;
    for (int k = tid; k < nbins; k += nthr) { s0 += pw[k]; s1 += pw[k] * (k * fstep); }
    s0 = wave_sum_f64(s0); s1 = wave_sum_f64(s1);
    if (lane == 0) { s_red[wv][0] = s0; s_red[wv][1] = s1; }
    __syncthreads();
    double tot = 0.0, cog = 0.0;
    for (int q = 0; q < nwv; ++q) { tot += s_red[q][0]; cog += s_red[q][1]; }
    cog /= tot;
    __syncthreads();
    double m2 = 0, m3 = 0, m4 = 0;
    for (int k = tid; k < nbins; k += nthr) {
        const double d = k * fstep - cog, p = pw[k];
        const double d2 = d * d;
        m2 += p * d2; m3 += p * d2 * d; m4 += p * d2 * d2;
    }
    m2 = wave_sum_f64(m2); m3 = wave_sum_f64(m3); m4 = wave_sum_f64(m4);
    if (lane == 0) { s_red[wv][0] = m2; s_red[wv][1] = m3; s_red[wv][2] = m4; }
    __syncthreads();
    if (tid == 0) {
        double u2 = 0.0, u3 = 0.0, u4 = 0.0;
        for (int q = 0; q < nwv; ++q) { u2 += s_red[q][0]; u3 += s_red[q][1]; u4 += s_red[q][2]; }
        u2 /= tot; u3 /= tot; u4 /= tot;
        o[0] = 1.0;
        o[1] = cog;
        o[2] = sqrt(u2);
        o[3] = u3 / (u2 * sqrt(u2));
        o[4] = u4 / (u2 * u2) - 3.0;
    }
}

// mean of each moment over gated frames whose value is not NaN (reference :366-374)
// The same analysis, one wavefront per frame, for the usual transform length of 1 024 (25 ms Gaussian window at 16 kHz):
// the real transform is one 512-point complex transform in registers (wave_fft.h), the power of bins k and 512 - k comes
// from the conjugate pair the lane holding k < 256 evaluates, and the four moments are sums over the registers.
constexpr int SPM_FRAMES = 4;
__global__ __launch_bounds__(64, 4) void spec_moments_wave_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                                  const ClipInfo* __restrict__ pitch_ci, const double* __restrict__ sel_freq,
                                                                  double pitch_dt, double ceiling, const double* __restrict__ win,
                                                                  const double2* __restrict__ tw, int nsamp, int half, int nbins,
                                                                  double tstep, double fstep, double* __restrict__ mom) {
    using namespace wfft;
    __shared__ double lds[Plan<8>::LDS_DOUBLES];
    constexpr int R = 8, S = 512, H = 4;
    const ClipInfo c = ci[blockIdx.y];
    const int f0 = blockIdx.x * SPM_FRAMES;
    if (f0 >= c.n_frames) return;
    const ClipInfo pc = pitch_ci[blockIdx.y];
    const int lane_ = threadIdx.x;
    const float* x = wav + c.sample_off;
    const int n = c.n_samples;
    LdsMem mem{lds};
    const int f1 = f0 + SPM_FRAMES < c.n_frames ? f0 + SPM_FRAMES : c.n_frames;
#pragma unroll 1
    for (int f = f0; f < f1; ++f) {
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        double* o = mom + (c.frame_off + f) * 5;
        const double t = c.t1 + f * tstep;
        // gate: Pitch "Get value at time" defined iff the nearest pitch frame is voiced
        {
            const double ireal = (t - pc.t1) / pitch_dt;
            const double il = floor(ireal);
            const int64_t near = (ireal - il < 0.5) ? (int64_t)il : (int64_t)il + 1;
            bool ok = near >= 0 && near < pc.n_frames;
            if (ok) {
                const double pf = sel_freq[pc.frame_off + near];
                ok = pf > 0.0 && pf < ceiling;
            }
            if (!ok) {                                            // uniform
                if (lane == 0) o[0] = 0.0;
                continue;
            }
        }
        const int start = (int)(low_index(t, c.x1) + 1 - half);
        cplx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int j = 2 * (lane + 64 * m);                    // samples 2 k, 2 k + 1 of the frame (loads on clamped indices)
            int i0 = start + j, i1 = i0 + 1;
            i0 = i0 < 0 ? 0 : (i0 > n - 1 ? n - 1 : i0);
            i1 = i1 < 0 ? 0 : (i1 > n - 1 ? n - 1 : i1);
            const int w0 = j < nsamp ? j : nsamp - 1, w1 = j + 1 < nsamp ? j + 1 : nsamp - 1;
            const double a0 = (double)x[i0] * win[w0], a1 = (double)x[i1] * win[w1];
            v[m] = cplx{j < nsamp ? a0 : 0.0, j + 1 < nsamp ? a1 : 0.0};
        }
        const double2_t wl2 = reinterpret_cast<const double2_t*>(tw)[lane];
        const cplx wl{wl2.x, wl2.y};
        {
            const double2_t a = reinterpret_cast<const double2_t*>(tw)[2 * lane], b = reinterpret_cast<const double2_t*>(tw)[(lane % 8) * 16];
            wave_fft<R>(v, lds, lane, cplx{a.x, a.y}, cplx{b.x, b.y});
        }
        ac_spec_store<R>(v, mem, lane);
        wave_sync();
        // power of bins k (pa) and 512 - k (pb) for the lane's k = lane + 64 m < 256; lane 0 also holds bin 256 (p256)
        double pa[H], pb[H];
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const int k = lane + 64 * m;
            const int pp = k ? S / 2 - k : 0;
            cplx zc{mem.ld(pp), mem.ld(S / 2 + pp)};
            if (m == 0) zc = cplx{lane == 0 ? v[0].x : zc.x, lane == 0 ? v[0].y : zc.y};
            const cplx zk = v[m], w = mul_w64(wl, m * 4);                       // W_1024^(64 m) = W_64^(4 m)
            const double er = 0.5 * (zk.x + zc.x), ei = 0.5 * (zk.y - zc.y), orr = 0.5 * (zk.y + zc.y), oi = -0.5 * (zk.x - zc.x);
            const double tr = w.x * orr - w.y * oi, ti = w.x * oi + w.y * orr;
            const double ar = er + tr, ai = ei + ti, br = er - tr, bi = ti - ei;
            pa[m] = ar * ar + ai * ai;
            pb[m] = br * br + bi * bi;
        }
        const double p256 = v[H].x * v[H].x + v[H].y * v[H].y;
        wave_sync();
        // bins: k (pa[m]), 512 - k (pb[m]; k = 0: bin 512), 256 (lane 0)
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const int k = lane + 64 * m, kb = S - k;
            if (k < nbins) { s0 += pa[m]; s1 += pa[m] * (k * fstep); }
            if (kb < nbins) { s0 += pb[m]; s1 += pb[m] * (kb * fstep); }
        }
        if (lane == 0 && S / 2 < nbins) { s0 += p256; s1 += p256 * ((S / 2) * fstep); }
        const double tot = group_sum<64>(s0);
        const double cog = group_sum<64>(s1) / tot;
        double m2 = 0.0, m3 = 0.0, m4 = 0.0;
        auto acc = [&](double p, int k) {
            const double d = k * fstep - cog, d2 = d * d;
            m2 += p * d2; m3 += p * d2 * d; m4 += p * d2 * d2;
        };
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const int k = lane + 64 * m, kb = S - k;
            if (k < nbins) acc(pa[m], k);
            if (kb < nbins) acc(pb[m], kb);
        }
        if (lane == 0 && S / 2 < nbins) acc(p256, S / 2);
        const double u2 = group_sum<64>(m2) / tot, u3 = group_sum<64>(m3) / tot, u4 = group_sum<64>(m4) / tot;
        if (lane == 0) {
            o[0] = 1.0;
            o[1] = cog;
            o[2] = sqrt(u2);
            o[3] = u3 / (u2 * sqrt(u2));
            o[4] = u4 / (u2 * u2) - 3.0;
        }
    }
}

__global__ __launch_bounds__(64) void moments_stats_kernel(const double* __restrict__ mom, const ClipInfo* __restrict__ ci,
                                                           double* __restrict__ out) {
    const ClipInfo c = ci[blockIdx.x];
    const int lane = threadIdx.x;
    double n[4] = {0, 0, 0, 0}, s[4] = {0, 0, 0, 0};
    for (int i = lane; i < c.n_frames; i += 64) {
        const double* m = mom + (c.frame_off + i) * 5;
        if (m[0] != 0.0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double v = m[1 + q];
                if (v == v) { n[q] += 1; s[q] += v; }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { n[q] = wave_sum_f64(n[q]); s[q] = wave_sum_f64(s[q]); }
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            out[blockIdx.x * 4 + q] = n[q] > 0 ? s[q] / n[q] : __longlong_as_double(0x7ff8000000000000LL);
}

// ---- _speechrate (de Jong & Wempe syllable nuclei; src/mshds_extractor.py:11-125) -----------------------
// One wave per clip.  Parallel parts: parabolic extrema, rank sort (0.99 quantile), local maxima and
// their sinc-70 refined times (four 16-lane groups).  The interval / peak bookkeeping is integer-state
// sequential logic and runs on lane 0 with a per-clip global workspace.
__device__ double value_cubic(const double* __restrict__ y, int n, double ireal) {
    const double x1 = ireal + 1.0;
    if (x1 > n) return y[n - 1];
    if (x1 < 1) return y[0];
    const int midleft = (int)floor(x1);
    if (x1 == (double)midleft) return y[midleft - 1];
    const int midright = midleft + 1;
    int depth = 2;
    if (depth > midright - 1) depth = midright - 1;
    if (depth > n - midleft) depth = n - midleft;
    if (depth <= 0) return y[(int)floor(x1 + 0.5) - 1];
    const double yl = y[midleft - 1], yr = y[midright - 1];
    if (depth == 1) return yl + (x1 - midleft) * (yr - yl);
    const double dyl = 0.5 * (yr - y[midleft - 2]), dyr = 0.5 * (y[midright] - yl);
    const double fil = x1 - midleft, fir = midright - x1;
    return yl * fir + yr * fil - fil * fir * (0.5 * (dyr - dyl) + (fil - 0.5) * (dyl + dyr - 2.0 * (yr - yl)));
}

constexpr int SR_MAX_PEAKS = 4096;

__global__ __launch_bounds__(64) void speechrate_kernel(const double* __restrict__ db_all, const ClipInfo* __restrict__ ci,
                                                        double dt, const double* __restrict__ sel_freq,
                                                        const ClipInfo* __restrict__ pci, double pitch_dt, double ceiling,
                                                        double* __restrict__ work, int64_t work_stride, int max_frames,
                                                        int peak_cap, int in_global, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const ClipInfo c = ci[blockIdx.x];
    const int n = c.n_frames, lane = threadIdx.x;
    double* o = out + (int64_t)blockIdx.x * 5;
    const double qn = __longlong_as_double(0x7ff8000000000000LL);
    if (n <= 0) {
        if (lane < 5) o[lane] = qn;
        return;
    }
    // contours that do not fit the LDS (clips beyond ~2 minutes) live in the per-clip global scratch instead
    double* gscr = work + (int64_t)blockIdx.x * work_stride + 3 * ((int64_t)max_frames + 2) + 2 * (int64_t)peak_cap;
    double* y = in_global ? gscr : reinterpret_cast<double*>(smem_raw);      // [n] intensity contour
    double* srt = y + ((n + 1) & ~1);                          // [n] sorted copy, later peak positions
    int* pk = reinterpret_cast<int*>(srt + ((n + 1) & ~1));    // [peak_cap] local-maximum indices
    const double* src = db_all + c.frame_off;
    for (int i = lane; i < n; i += 64) y[i] = src[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // parabolic extrema (Vector_getMaximum / Minimum)
    double mx = -INFINITY, mnn = -INFINITY;
    for (int i = lane; i < n; i += 64) {
        const double v = y[i];
        if (i == 0 || i == n - 1) { mx = fmax(mx, v); mnn = fmax(mnn, -v); }
        if (i > 0 && i < n - 1) {
            const double a = y[i - 1], b = y[i + 1];
            if (v > a && v >= b) { const double dy = 0.5 * (b - a), d2 = 2.0 * v - a - b; mx = fmax(mx, d2 != 0.0 ? v + 0.5 * dy * dy / d2 : v); }
            if (-v > -a && -v >= -b) { const double dy = 0.5 * (a - b), d2 = -2.0 * v + a + b; mnn = fmax(mnn, d2 != 0.0 ? -v + 0.5 * dy * dy / d2 : -v); }
        }
    }
    const double max_int = wave_max_f64(mx), min_int = -wave_max_f64(mnn);
    // rank sort -> 0.99 quantile (Praat NUMquantile)
    for (int i = lane; i < n; i += 64) {
        const double v = y[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) { const double u = y[j]; rank += (u < v) || (u == v && j < i); }
        srt[rank] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double q99;
    if (n == 1) q99 = srt[0];
    else {
        const double place = 0.99 * n + 0.5;
        int left = (int)floor(place);
        left = left < 1 ? 1 : (left > n - 1 ? n - 1 : left);
        q99 = (srt[left] == srt[left - 1]) ? srt[left - 1] : srt[left - 1] + (place - left) * (srt[left] - srt[left - 1]);
    }
    const double silencedb = -25.0, mindip = 2.0, minpause = 0.3, minsound = 0.1;
    double silencedb_1 = q99 + silencedb;
    if (silencedb_1 < min_int) silencedb_1 = min_int;
    const double silencedb_2 = silencedb - (max_int - q99);
    // local maxima in ascending order
    int npk = 0;
    for (int base = 1; base < n - 1; base += 64) {
        const int i = base + lane;
        bool ok = false;
        if (i < n - 1) ok = y[i] > y[i - 1] && y[i] >= y[i + 1];
        const unsigned long long m = __ballot(ok);
        const int pos = npk + __popcll(m & ((1ull << lane) - 1ull));
        if (ok && pos < peak_cap) pk[pos] = i;
        npk += __popcll(m);
    }
    if (npk > peak_cap) npk = peak_cap;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // sinc-70 refined peak positions (reuse srt[] for them), four peaks at a time
    {
        const int l16 = lane & 15, grp = lane >> 4;
        for (int b = 0; b < npk; b += 4) {
            const int k = b + grp;
            const bool live = k < npk;
            double xm, ym;
            improve_max_group<16, false>(y, n, (double)pk[live ? k : 0], 70, 0, n - 1, l16, live, xm, ym);
            if (live && l16 == 0) srt[k] = xm;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane != 0) return;
    // ------------------------------ lane 0: sequential bookkeeping ------------------------------
    double* iva = work + (int64_t)blockIdx.x * work_stride;    // interval starts
    double* ivb = iva + (n + 2);                               // interval ends
    double* ivl = ivb + (n + 2);                               // 1 = sounding, 0 = silent
    double* tpk = ivl + (n + 2);                               // kept peak times
    double* vpk = tpk + peak_cap;                              // kept peak values
    const double duration = c.xmax;                            // the Intensity (and its TextGrid) keep the sound's domain [0, xmax]
    int niv = 0;
    {
        const double thr = max_int - fabs(silencedb_2);
        if (minpause > duration || thr < min_int) {
            iva[0] = 0.0; ivb[0] = duration; ivl[0] = 1.0; niv = 1;
        } else {
            double start = 0.0;
            bool state = y[0] < thr;                            // in silence
            for (int i = 1; i < n; ++i) {
                const bool sil = y[i] < thr;
                if (sil != state) {
                    const double tb = c.t1 + (i - 0.5) * dt;
                    iva[niv] = start; ivb[niv] = tb; ivl[niv] = state ? 0.0 : 1.0; ++niv;
                    start = tb; state = sil;
                }
            }
            iva[niv] = start; ivb[niv] = duration; ivl[niv] = state ? 0.0 : 1.0; ++niv;
            for (int pass = 0; pass < 2; ++pass) {
                const double lab = pass == 0 ? 1.0 : 0.0;       // cut short sounding first, then short silences
                const double mind = pass == 0 ? minsound : minpause;
                int i = 0;
                while (i < niv) {
                    if (ivl[i] == lab && (ivb[i] - iva[i]) < mind && niv > 1) {
                        const double a = iva[i], b = ivb[i];
                        for (int k = i; k < niv - 1; ++k) { iva[k] = iva[k + 1]; ivb[k] = ivb[k + 1]; ivl[k] = ivl[k + 1]; }
                        --niv;
                        if (i == 0) iva[0] = a;
                        else ivb[i - 1] = b;                     // previous interval extended (also for the last one)
                    } else ++i;
                }
                const double ml = pass == 0 ? 0.0 : 1.0;        // merge equal neighbours of the other label
                i = 0;
                while (i < niv - 1) {
                    if (ivl[i] == ml && ivl[i + 1] == ivl[i]) {
                        const double a = iva[i];
                        for (int k = i; k < niv - 1; ++k) { iva[k] = iva[k + 1]; ivb[k] = ivb[k + 1]; ivl[k] = ivl[k + 1]; }
                        --niv;
                        iva[i] = a;
                    } else ++i;
                }
            }
        }
    }
    int npauses = 0;
    double phonation = 0.0, begin_speak = 0.0, end_speak = 0.0;
    for (int i = 0; i < niv; ++i)
        if (ivl[i] != 0.0) {
            if (npauses == 0) begin_speak = iva[i];
            end_speak = ivb[i];
            phonation += ivb[i] - iva[i];
            ++npauses;
        }
    if (npauses == 0) {
        for (int k = 0; k < 5; ++k) o[k] = qn;
        return;
    }
    int nkeep = 0;
    for (int k = 0; k < npk; ++k) {
        const double v = value_cubic(y, n, srt[k]);
        if (v > silencedb_1) { tpk[nkeep] = c.t1 + srt[k] * dt; vpk[nkeep] = v; ++nkeep; }
    }
    const ClipInfo pc = pci[blockIdx.x];
    int nsyll = 0;
    if (nkeep > 1) {
        double currenttime = tpk[0], currentint = vpk[0];
        for (int p = 0; p < nkeep - 1; ++p) {
            const double nxt = tpk[p + 1];
            int imin = (int)ceil((currenttime - c.t1) / dt), imax = (int)floor((nxt - c.t1) / dt);
            imin = imin < 0 ? 0 : imin;
            imax = imax > n - 1 ? n - 1 : imax;
            double dip;
            if (imin <= imax) {
                dip = y[imin];
                for (int i = imin + 1; i <= imax; ++i) dip = fmin(dip, y[i]);
            } else {
                int ia = (int)floor((currenttime - c.t1) / dt + 0.5), ib = (int)floor((nxt - c.t1) / dt + 0.5);
                ia = ia < 0 ? 0 : (ia > n - 1 ? n - 1 : ia);
                ib = ib < 0 ? 0 : (ib > n - 1 ? n - 1 : ib);
                dip = fmin(y[ia], y[ib]);
            }
            if (fabs(currentint - dip) > mindip) {
                // valid syllable nucleus at tpk[p]: count it if it is in a sounding interval and voiced
                const double tm = tpk[p];
                bool snd = false;
                for (int k = 0; k < niv; ++k)
                    if ((iva[k] <= tm && tm < ivb[k]) || (k == niv - 1 && tm == ivb[k])) { snd = ivl[k] != 0.0; break; }
                bool voiced = false;
                if (pc.n_frames > 0) {
                    const double ireal = (tm - pc.t1) / pitch_dt;
                    const double il = floor(ireal);
                    const int64_t near = (ireal - il < 0.5) ? (int64_t)il : (int64_t)il + 1;
                    if (near >= 0 && near < pc.n_frames) {
                        const double pf = sel_freq[pc.frame_off + near];
                        voiced = pf > 0.0 && pf < ceiling;
                    }
                }
                if (snd && voiced) ++nsyll;
            }
            currenttime = nxt;
            currentint = value_cubic(y, n, (nxt - c.t1) / dt);
        }
    }
    const double original_dur = end_speak - begin_speak;
    const int n_pauses = npauses - 1;
    o[0] = original_dur > 0 ? nsyll / original_dur : 0.0;
    o[1] = phonation > 0 ? nsyll / phonation : 0.0;
    o[2] = original_dur > 0 ? phonation / original_dur : 0.0;
    o[3] = original_dur > 0 ? n_pauses / original_dur : 0.0;
    o[4] = n_pauses > 0 ? (original_dur - phonation) / n_pauses : 0.0;
}

// ---- 16 kHz -> 10 kHz resampling (Sound_resample (10000, 500) at the head of To Formant (burg)) -----------------
// Input: the clip after Praat's FFT low-pass (rsaf_praat_lowpass_batch).  out[m] = sum_k x[base + k] * W[phase][k + D]:
// the ratio 5/8 gives 5 distinct fractional offsets, whose NUM_interpolate_sinc weights at full depth D the host
// tabulates in float64.
struct ResampleInfo {        // per clip (host-built), 48 bytes
    int64_t sample_off;      // into wav
    int64_t out_off;         // into the 10 kHz buffer
    double pos0;             // real input index of output sample 0
    double x1o;              // time of output sample 0
    int n_in, n_out;
    int table;               // index of this clip's weight table (one per distinct pos0)
    int pad;
};

// 320-thread workgroup = 5 phases x 256 consecutive q: wave r owns phase r, so its weight row is wave-uniform and comes
// through the scalar cache into SGPRs (no LDS traffic for the weights); a lane owns 4 consecutive q, whose tap windows
// are 8 samples apart: every sample it reads from the LDS tile feeds 4 FMAs (taps k, k - 8, k - 16, k - 24 of its four
// outputs), which balances the LDS read rate against the fp64 FMA rate.  Lanes are 32 samples apart in the tile; a
// 33/32 skew puts the 8-byte reads of a half wave on distinct banks.  The tables hold NUM_interpolate_sinc's weights at
// full depth for the five fractional positions of the 8 : 5 grid, rows zero-padded to `wstride` doubles; outputs whose
// depth Praat clips (within `depth` input samples of either end) are recomputed by resample_edge_kernel.
constexpr int RS_QL = 4;                  // consecutive q per lane
constexpr int RS_QT = 64 * RS_QL;         // q per workgroup
__global__ __launch_bounds__(320) void resample_kernel(const double* __restrict__ lp, const ResampleInfo* __restrict__ ri,
                                                       const double* __restrict__ tables, int wstride,
                                                       const int* __restrict__ phase_base, int depth,
                                                       double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const ResampleInfo c = ri[blockIdx.y];
    const int q0 = blockIdx.x * RS_QT;
    if (5 * q0 >= c.n_out) return;
    const int taps = 2 * depth + 1;
    double* xs = reinterpret_cast<double*>(smem_raw);                 // input tile
    const int tid = threadIdx.x;
    const int* pb = phase_base + c.table * 5;
    int bmin = pb[0], bmax = pb[0];
    for (int r = 1; r < 5; ++r) { bmin = min(bmin, pb[r]); bmax = max(bmax, pb[r]); }
    const int lo = 8 * q0 + bmin - depth;                             // first input index of the tile
    const int nkb = (taps + 8 * (RS_QL - 1) + 7) / 8;                 // tap blocks of 8; the rows are zero beyond `taps`
    const int span = 8 * RS_QL * 63 + (bmax - bmin) + 8 * nkb;        // every index the tap loop reads
    const double* x = lp + c.sample_off;
    for (int i = tid; i < span; i += 320) {
        const int j = lo + i;
        xs[i + (i >> 5)] = (j >= 0 && j < c.n_in) ? x[j] : 0.0;
    }
    __syncthreads();
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6), ql = tid & 63;
    const double* __restrict__ w = tables + ((int64_t)c.table * 5 + r) * wstride;   // wave-uniform
    const int i0 = 8 * RS_QL * ql + pb[r] - bmin;                     // tile index of tap 0 of the lane's first output
    double acc[RS_QL] = {0.0, 0.0, 0.0, 0.0};
    double wq[RS_QL][8];                                              // weights k = 8 (kb - j) + t of output j
#pragma unroll
    for (int j = 0; j < RS_QL; ++j)
#pragma unroll
        for (int t = 0; t < 8; ++t) wq[j][t] = 0.0;
    for (int kb = 0; kb < nkb; ++kb) {
#pragma unroll
        for (int j = RS_QL - 1; j > 0; --j)
#pragma unroll
            for (int t = 0; t < 8; ++t) wq[j][t] = wq[j - 1][t];
#pragma unroll
        for (int t = 0; t < 8; ++t) wq[0][t] = w[8 * kb + t];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = i0 + 8 * kb + t;
            const double xv = xs[i + (i >> 5)];
#pragma unroll
            for (int j = 0; j < RS_QL; ++j) acc[j] = fma(xv, wq[j][t], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < RS_QL; ++j) {
        const int m = 5 * (q0 + RS_QL * ql + j) + r;
        if (m < c.n_out) out[c.out_off + m] = acc[j];
    }
}

// the first and last `n_edge` output samples of every clip by the general routine when their depth is clipped (or the
// position falls outside the sound)
__global__ __launch_bounds__(256) void resample_edge_kernel(const double* __restrict__ lp, const ResampleInfo* __restrict__ ri,
                                                            int depth, int n_edge, double ratio_in_out, double* __restrict__ out) {
    const ResampleInfo c = ri[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 2 * n_edge) return;
    const int m = e < n_edge ? e : c.n_out - 1 - (e - n_edge);
    if (m < 0 || m >= c.n_out) return;
    const double x = c.pos0 + (double)m * ratio_in_out + 1.0;         // Praat's 1-based real index
    const int64_t midleft = (int64_t)floor(x);
    const bool full = x >= 1.0 && x <= (double)c.n_in && midleft >= depth && (int64_t)c.n_in - midleft >= depth;
    if (full) return;
    out[c.out_off + m] = praat_interpolate_sinc(lp + c.sample_off, (int64_t)c.n_in, x, depth);
}

// ---- Formant (burg): one wave per frame -----------------------------------------------------------------------
// (wave sums / maxima on DPP + v_readlane, root broadcasts on v_readlane: the ds_bpermute forms - ~50 LDS-crossbar round
// trips per Aberth iteration, 24 per Burg order - were most of the 34 000 cycles a frame took)
// Gaussian-windowed 50 ms frame of the pre-emphasised 10 kHz signal -> Burg LPC (order 10) -> roots by
// Aberth-Ehrlich iteration (all ten simultaneously, lanes 0..9) + Newton polish -> reflect into the unit
// circle -> (frequency, bandwidth) of the roots in the upper half plane, ascending, at most 5.
constexpr int FB_ORDER = 10;
struct FormantFrame { double f[5]; double b[5]; };

// A wave takes FB_GROUP consecutive frames: the Burg recursion runs frame by frame with all 64 lanes on the frame's
// samples; the root finding - ten lanes per polynomial - then runs for the FB_GROUP polynomials at once (lane = 10 g + root),
// so that the most expensive phase (~650 instructions per Aberth iteration, 10-15 iterations) is paid once per 6 frames
// instead of once per frame with 54 idle lanes.
constexpr int FB_GROUP = 6;
__global__ __launch_bounds__(256) void formant_kernel(const double* __restrict__ y10, const ResampleInfo* __restrict__ ri,
                                                      const ClipInfo* __restrict__ ci, const double* __restrict__ win,
                                                      int nsw, double dt, double dxo, double preemph,
                                                      FormantFrame* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ double s_cf[4][FB_GROUP][FB_ORDER + 1];
    __shared__ double2 s_z[4][64];
    __shared__ double s_fq[4][64];
    __shared__ int s_okf[4][FB_GROUP];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ClipInfo c = ci[blockIdx.y];
    const int fbase = (blockIdx.x * 4 + wv) * FB_GROUP;
    if (fbase >= c.n_frames) return;
    const ResampleInfo r = ri[blockIdx.y];
    double* b1 = reinterpret_cast<double*>(smem_raw) + (size_t)wv * 2 * (nsw + 2);
    double* b2 = b1 + (nsw + 2);
    const double* y = y10 + r.out_off;
    const int n = r.n_out;
    const double x1o = r.x1o;
    const double qn = __longlong_as_double(0x7ff8000000000000LL);
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const int ng = min(FB_GROUP, c.n_frames - fbase);
#pragma unroll 1
    for (int g = 0; g < ng; ++g) {
        const int f = fbase + g;
        const double t = c.t1 + f * dt;
        const int left = (int)floor((t - x1o) / dxo);
        const int half = nsw / 2;
        int start = left + 1 - half, end = left + half;
        start = start < 0 ? 0 : start;
        end = end > n - 1 ? n - 1 : end;
        const int len = end - start + 1;
        // pre-emphasised, windowed frame into b1[1..len] (Burg's 1-based arrays); also the max intensity
        double mxi = 0.0, p = 0.0;
        for (int j0 = lane; j0 < len; j0 += 4 * 64) {                     // twelve loads in flight (see clip_peak_kernel)
            double ya[4], yb[4], wq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 64 * u < len ? j0 + 64 * u : len - 1, i = start + j;
                ya[u] = y[i];
                yb[u] = y[i > 0 ? i - 1 : 0];
                wq[u] = win[j];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 64 * u, i = start + j;
                if (j < len) {
                    const double v = (i > 0) ? ya[u] - preemph * yb[u] : ya[u];
                    mxi = fmax(mxi, v * v);
                    const double xv = v * wq[u];
                    b1[j + 1] = xv;
                    p += xv * xv;
                }
            }
        }
        mxi = wave_max_dpp(mxi);
        p = group_sum<64>(p);
        bool ok = !(len < FB_ORDER + 2 || mxi == 0.0 || p <= 0.0);
        wsync();
        double a[FB_ORDER + 1], aa[FB_ORDER + 1];
#pragma unroll
        for (int i = 0; i <= FB_ORDER; ++i) { a[i] = 0.0; aa[i] = 0.0; }
        if (ok) {
            // NUMburg.  x = b1 copy: b2[j] = x[j+1] for j = 1..len-1, b1[j] = x[j] for j = 1..len-1
            for (int j = 1 + lane; j <= len - 1; j += 64) b2[j] = b1[j + 1];
            wsync();
            for (int i = 1; i <= FB_ORDER; ++i) {
                double num = 0.0, den = 0.0;
                for (int j = 1 + lane; j <= len - i; j += 64) { const double u = b1[j], v = b2[j]; num += u * v; den += u * u + v * v; }
                num = group_sum<64>(num);
                den = group_sum<64>(den);
                if (den <= 0.0) { ok = false; break; }
                a[i] = 2.0 * num / den;
                for (int j = 1; j < i; ++j) a[j] = aa[j] - a[i] * aa[i - j];
                if (i < FB_ORDER) {
                    for (int j = 1; j <= i; ++j) aa[j] = a[j];
                    const double k = aa[i];
                    // b1[j] -= k*b2[j]; b2[j] = b2[j+1] - k*b1[j+1] (old b1) for j = 1..len-i-1
                    for (int j0 = 1; j0 <= len - i - 1; j0 += 64) {
                        const int j = j0 + lane;
                        double nb1 = 0.0, nb2 = 0.0;
                        const bool on = j <= len - i - 1;
                        if (on) { nb1 = b1[j] - k * b2[j]; nb2 = b2[j + 1] - k * b1[j + 1]; }
                        __builtin_amdgcn_wave_barrier();
                        if (on) { b1[j] = nb1; b2[j] = nb2; }
                        wsync();
                    }
                }
            }
        }
        // polynomial z^10 - a1 z^9 - ... - a10 ; cf[k] = coefficient of z^(10-k)
        if (lane == 0) {
            s_okf[wv][g] = ok ? 1 : 0;
            s_cf[wv][g][0] = 1.0;
#pragma unroll
            for (int k = 1; k <= FB_ORDER; ++k) s_cf[wv][g][k] = -a[k];
        }
        wsync();
    }
    // ---- roots of the ng polynomials at once: lane = 10 g + root ----
    const int g = lane / FB_ORDER, li = lane - g * FB_ORDER;
    const bool mine = g < ng;
    const int gg = mine ? g : 0;
    const bool okf = mine && s_okf[wv][gg] != 0;
    double cf[FB_ORDER + 1];
#pragma unroll
    for (int k = 0; k <= FB_ORDER; ++k) cf[k] = s_cf[wv][gg][k];
    // Aberth-Ehrlich: start on a circle of radius 0.9
    double zr = 0.0, zi = 0.0;
    { double sn, cs; sincos(2.0 * PI * (li + 0.35) / FB_ORDER, &sn, &cs); zr = 0.9 * cs; zi = 0.9 * sn; }
    for (int it = 0; it < 80; ++it) {
        // p(z), p'(z) by Horner
        double pr = cf[0], pi_ = 0.0, dr = 0.0, di = 0.0;
#pragma unroll
        for (int k = 1; k <= FB_ORDER; ++k) {
            const double ndr = dr * zr - di * zi + pr, ndi = dr * zi + di * zr + pi_;
            dr = ndr; di = ndi;
            const double npr = pr * zr - pi_ * zi + cf[k], npi = pr * zi + pi_ * zr;
            pr = npr; pi_ = npi;
        }
        // w = p/p'
        const double dd = dr * dr + di * di;
        double wr_ = 0.0, wi_ = 0.0;
        if (dd > 0.0) { const double rd = fast_rcp(dd); wr_ = (pr * dr + pi_ * di) * rd; wi_ = (pi_ * dr - pr * di) * rd; }
        // s = sum_{j != i} 1/(z_i - z_j) over the roots of the same polynomial
        s_z[wv][lane] = make_double2(zr, zi);
        wsync();
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int j = 0; j < FB_ORDER; ++j) {
            const double2 oz = s_z[wv][gg * FB_ORDER + j];
            const double ex = zr - oz.x, ey = zi - oz.y;
            const double ee = ex * ex + ey * ey;
            if (j != li && ee > 0.0) { const double re = fast_rcp(ee); sr += ex * re; si -= ey * re; }
        }
        wsync();
        // delta = w / (1 - w*s)
        const double qr = 1.0 - (wr_ * sr - wi_ * si), qi = -(wr_ * si + wi_ * sr);
        const double qq = qr * qr + qi * qi;
        double er = wr_, ei = wi_;
        if (qq > 0.0) { const double rq = fast_rcp(qq); er = (wr_ * qr + wi_ * qi) * rq; ei = (wi_ * qr - wr_ * qi) * rq; }
        zr -= er; zi -= ei;
        const double step = okf ? fabs(er) + fabs(ei) : 0.0;
        if (wave_max_dpp(step) < 1e-11) break;               // the three Newton steps below square this down to rounding
    }
    for (int it = 0; it < 3; ++it) {                              // Newton polish on the original polynomial
        double pr = cf[0], pi_ = 0.0, dr = 0.0, di = 0.0;
#pragma unroll
        for (int k = 1; k <= FB_ORDER; ++k) {
            const double ndr = dr * zr - di * zi + pr, ndi = dr * zi + di * zr + pi_;
            dr = ndr; di = ndi;
            const double npr = pr * zr - pi_ * zi + cf[k], npi = pr * zi + pi_ * zr;
            pr = npr; pi_ = npi;
        }
        const double dd = dr * dr + di * di;
        if (dd > 0.0) { zr -= (pr * dr + pi_ * di) / dd; zi -= (pi_ * dr - pr * di) / dd; }
    }
    // fix into the unit circle, keep the upper half plane, convert
    const double nyq = 0.5 / dxo;
    double mag2 = zr * zr + zi * zi;
    if (mag2 > 1.0) { zr /= mag2; zi /= mag2; mag2 = zr * zr + zi * zi; }   // z -> 1/conj(z)
    double fq = fabs(atan2(zi, zr)) * nyq / PI;
    const double bw = -log(mag2) * nyq / PI;
    const bool keep = okf && zi >= 0.0 && fq >= 50.0 && fq <= nyq - 50.0;
    if (!keep) fq = 1e300;
    // rank among the kept roots of the same frame (stable by root index), write the first five
    s_fq[wv][lane] = fq;
    wsync();
    int rank = 0;
#pragma unroll
    for (int j = 0; j < FB_ORDER; ++j) {
        const double of = s_fq[wv][gg * FB_ORDER + j];
        rank += (of < fq) || (of == fq && j < li);
    }
    // the five lowest of each frame through LDS, so that one lane writes each output slot
    s_z[wv][lane] = make_double2(qn, qn);
    wsync();
    if (keep && rank < 5) s_z[wv][gg * FB_ORDER + rank] = make_double2(fq, bw);
    wsync();
    if (mine && li < 5) {
        FormantFrame* o = out + c.frame_off + fbase + gg;
        const double2 v = s_z[wv][gg * FB_ORDER + li];
        o->f[li] = v.x;
        o->b[li] = v.y;
    }
}

// ---- glottal pulses: Sound & Pitch: To PointProcess (cc), one wave per clip ----------------------------------
__device__ double pitch_value_at(const double* __restrict__ f, int n, double t1, double dt, double ceiling, double t) {
    const double qn = __longlong_as_double(0x7ff8000000000000LL);
    if (n <= 0) return qn;
    const double ireal = (t - t1) / dt;
    const int64_t ileft = (int64_t)floor(ireal);
    double phase = ireal - (double)ileft;
    int64_t inear, ifar;
    if (phase < 0.5) { inear = ileft; ifar = ileft + 1; } else { inear = ileft + 1; ifar = ileft; phase = 1.0 - phase; }
    if (inear < 0 || inear >= n) return qn;
    const double fn = f[inear];
    if (!(fn > 0.0 && fn < ceiling)) return qn;
    if (ifar < 0 || ifar >= n) return fn;
    const double ff = f[ifar];
    if (!(ff > 0.0 && ff < ceiling)) return fn;
    return fn + phase * (ff - fn);
}

// Sound_findMaximumCorrelation with the shifts spread over the lanes; returns corr, *tout, *peak (uniform).
// The samples come from a wave-private LDS window that SLIDES with the walk: the fixed window and the union of the
// shifted windows of one pulse span ~2.5 periods, the window holds PULSE_LDS samples, so it is refilled from global
// memory once per ~10-20 pulses (in the walking direction) instead of twice per pulse; every lane then reads the fixed
// window as a broadcast and its own shifted window with unit stride.  `win0` = sample index of the window's first
// entry (INT64_MIN: empty), kept by the caller across pulses; dir = -1 / +1: the walk goes left / right.
constexpr int PULSE_LDS = 3072;
constexpr int PULSE_FW = 128;         // pitch frames of the walker's sliding window
__device__ double max_correlation_wave(const float* __restrict__ x, int n, double x1, double t1, double window, double tmin2,
                                       double tmax2, int lane, double* tout, double* peak, float* pwin, int64_t* win0, int dir) {
    const double half = 0.5 * window;
    const int64_t ileft1 = nearest_index(t1 - half, x1);
    const int64_t iright1 = nearest_index(t1 + half, x1);
    const int64_t l2min = low_index(tmin2 - half, x1);
    const int64_t l2max = high_index(tmax2 - half, x1);
    double best = -1.0, r1 = 0.0, r2 = 0.0, r3 = 0.0, r1b = 0.0, r3b = 0.0, ir = 0.0, pk = 0.0;
    const int wlen = (int)(iright1 - ileft1 + 1);
    const int slen = (int)(l2max - l2min) + wlen;
    const int64_t ulo = ileft1 < l2min ? ileft1 : l2min;                      // union of both ranges
    const int64_t uhi = (ileft1 + wlen > l2min + slen ? ileft1 + wlen : l2min + slen);
    const bool staged = wlen > 0 && slen > 0 && uhi - ulo <= PULSE_LDS;
    if (staged && (*win0 == INT64_MIN || ulo < *win0 || uhi > *win0 + PULSE_LDS)) {
        // refill: the needed span at the trailing end of the window, the rest ahead in the walking direction
        const int64_t w0 = dir < 0 ? uhi - PULSE_LDS : ulo;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < PULSE_LDS; i += 64) { const int64_t j = w0 + i; pwin[i] = (j >= 0 && j < n) ? x[j] : 0.0f; }
        *win0 = w0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const float* ps1 = pwin + (staged ? (int)(ileft1 - *win0) : 0);
    const float* ps2 = pwin + (staged ? (int)(l2min - *win0) : 0);
    // Interior case (every sample of both windows lies inside the sound: all pulses but the ones at the very ends of a
    // clip): no pair is skipped, so the sum of squares of the fixed window is one number, the one of the shifted window
    // slides (norm2(s + 1) = norm2(s) - a[s]^2 + a[s + wlen]^2: one scan over the 64 shifts of a batch), and the local
    // peak is only needed for the one step that detects the maximum.  The loop over the window then carries the cross
    // product alone: 2 LDS reads, 2 conversions and 1 FMA per sample instead of 3 FMAs, a maximum and an absolute value more.
    const bool interior = staged && ulo >= 0 && uhi <= n;
    double n1_all = 0.0;
    if (interior) {
        for (int i = lane; i < wlen; i += 64) { const double a = ps1[i]; n1_all = fma(a, a, n1_all); }
        n1_all = group_sum<64>(n1_all);
    }
    for (int64_t b = l2min; b <= l2max; b += 64) {
        const int64_t ileft2 = b + lane;
        double norm1 = 0.0, norm2 = 0.0, prod = 0.0, lp = 0.0;
        if (interior) {
            const int ob = (int)(b - l2min);                       // window offset of the batch's first shift
            double n20 = 0.0;                                      // sum of squares of shift ob
            for (int i = lane; i < wlen; i += 64) { const double a = ps2[ob + i]; n20 = fma(a, a, n20); }
            n20 = group_sum<64>(n20);
            const bool in = ileft2 <= l2max;
            const int o2 = ob + lane;
            // d_t = a[t + wlen]^2 - a[t]^2 for shift t -> t + 1 (reads stay inside the union: the last lane that matters is cnt - 1)
            double dsc = 0.0;
            if (in && ileft2 < l2max) { const double lo_ = ps2[o2], hi_ = ps2[o2 + wlen]; dsc = hi_ * hi_ - lo_ * lo_; }
            double incl = dsc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const double up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
            norm1 = n1_all;
            norm2 = n20 + (incl - dsc);                            // exclusive prefix of the differences
            // The sliding sum is exact for 16-bit PCM (the squares add exactly); on resampled or float clips it can cancel
            // when the energy drops sharply inside the batch.  A lane whose sum has lost its digits takes the direct sum.
            if (in && !(norm2 > 1e-9 * n20)) {
                double d2 = 0.0;
                for (int i = 0; i < wlen; ++i) { const double a = ps2[o2 + i]; d2 = fma(a, a, d2); }
                norm2 = d2;
            }
            if (in) {
                double p0 = 0.0, p1 = 0.0;
                int i = 0;
                for (; i + 1 < wlen; i += 2) {
                    p0 = fma((double)ps1[i], (double)ps2[o2 + i], p0);
                    p1 = fma((double)ps1[i + 1], (double)ps2[o2 + i + 1], p1);
                }
                if (i < wlen) p0 = fma((double)ps1[i], (double)ps2[o2 + i], p0);
                prod = p0 + p1;
            }
        } else if (ileft2 <= l2max) {
            if (staged) {
                const int o2 = (int)(ileft2 - l2min);
                // Praat skips pairs outside the sound: the pairs inside are one index range, worked out once per lag
                int64_t lo = -ileft1 > -ileft2 ? -ileft1 : -ileft2, hi = n - ileft1 < n - ileft2 ? n - ileft1 : n - ileft2;
                lo = lo < 0 ? 0 : lo;
                hi = hi > wlen ? wlen : hi;
                for (int i = (int)lo; i < (int)hi; ++i) {
                    const double a1 = ps1[i], a2 = ps2[o2 + i];
                    norm1 += a1 * a1; norm2 += a2 * a2; prod += a1 * a2;
                    lp = fmax(lp, fabs(a2));
                }
            } else {
                for (int64_t i1 = ileft1, i2 = ileft2; i1 <= iright1; ++i1, ++i2) {
                    if (i1 < 0 || i1 >= n || i2 < 0 || i2 >= n) continue;
                    const double a1 = x[i1], a2 = x[i2];
                    norm1 += a1 * a1; norm2 += a2 * a2; prod += a1 * a2;
                    lp = fmax(lp, fabs(a2));
                }
            }
        }
        const double rr = prod != 0.0 ? prod / sqrt(norm1 * norm2) : 0.0;
        const int cnt = (int)((l2max - b + 1) < 64 ? (l2max - b + 1) : 64);
        // Praat's scan (r1 = r2; r2 = r3; r3 = r[k]; a strictly better r2 that is >= both neighbours wins, the local peak
        // taken at the step that detects it) for the 64 shifts at once: lane k holds step k's (r1, r2, r3) = (r[k-2], r[k-1],
        // r[k]) (r2, r3 carry the two last values across batches, zeros in front of the first shift), the winner is the FIRST
        // lane whose r2 equals the maximum over the qualifying lanes, taken only if it beats the best so far
        const double up1 = __shfl_up(rr, 1, 64), up2 = __shfl_up(rr, 2, 64);
        const double s2 = lane == 0 ? r3 : up1;
        const double s1 = lane == 0 ? r2 : (lane == 1 ? r3 : up2);
        const bool ok = lane < cnt && s2 >= s1 && s2 >= rr;
        const double m = wave_max_dpp(ok ? s2 : -INFINITY);
        if (m > best) {
            const int kw = __ffsll((long long)__ballot(ok && s2 == m)) - 1;
            best = m;
            r1b = readlane_f64(s1, kw); r3b = readlane_f64(rr, kw);
            if (interior) {                                        // local peak of the detecting step's shifted window
                const int ok_ = (int)(b - l2min) + kw;
                double mx = 0.0;
                for (int i = lane; i < wlen; i += 64) mx = fmax(mx, fabs((double)ps2[ok_ + i]));
                pk = wave_max_dpp(mx);
            } else {
                pk = readlane_f64(lp, kw);
            }
            ir = (double)(b + kw - 1);
        }
        const double last = readlane_f64(rr, cnt - 1);
        r2 = cnt >= 2 ? readlane_f64(rr, cnt - 2) : r3;
        r3 = last;
    }
    (void)r1;
    *peak = pk;
    *tout = t1;
    if (best > -1.0) {
        const double d2r = 2.0 * best - r1b - r3b;
        if (d2r != 0.0) { const double dr = 0.5 * (r3b - r1b); best += 0.5 * dr * dr / d2r; ir += dr / d2r; }
        *tout = t1 + (ir - (double)ileft1) * DXS;
    }
    return best;
}

__device__ double find_extremum_wave(const float* __restrict__ x, int n, double x1, double tmin, double tmax, int lane) {
    int64_t imin = low_index(tmin, x1), imax = high_index(tmax, x1);
    imin = imin < 0 ? 0 : imin;
    imax = imax > n - 1 ? n - 1 : imax;
    const int cnt = (int)(imax - imin + 1);
    if (cnt <= 0) return 0.5 * (tmin + tmax);
    double ie;
    if (cnt == 1) ie = 1.0;
    else if (cnt == 2) {
        const double a = fabs((double)x[imin]), b = fabs((double)x[imin + 1]);
        ie = a > b ? 1.0 : (a < b ? 2.0 : 1.5);
    } else {
        // first minimum / first maximum (strict comparisons in index order) via (value, index) reductions
        double mn = INFINITY, mx = -INFINITY;
        int jmn = 0x7fffffff, jmx = 0x7fffffff;
        for (int j = lane; j < cnt; j += 64) {
            const double v = x[imin + j];
            if (v < mn) { mn = v; jmn = j; }
            if (v > mx) { mx = v; jmx = j; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double omn = __shfl_xor(mn, o, 64), omx = __shfl_xor(mx, o, 64);
            const int ojmn = __shfl_xor(jmn, o, 64), ojmx = __shfl_xor(jmx, o, 64);
            if (omn < mn || (omn == mn && ojmn < jmn)) { mn = omn; jmn = ojmn; }
            if (omx > mx || (omx == mx && ojmx < jmx)) { mx = omx; jmx = ojmx; }
        }
        if (mn == mx) ie = 0.5 * (cnt + 1.0);
        else {
            const int j = fabs(mn) > fabs(mx) ? jmn : jmx;
            if (j == 0) ie = 1.0;
            else if (j == cnt - 1) ie = (double)cnt;
            else {
                const double vm = x[imin + j], vl = x[imin + j - 1], vr = x[imin + j + 1];
                ie = (j + 1) + 0.5 * (vr - vl) / (2.0 * vm - vl - vr);
            }
        }
    }
    return x1 + ((double)imin + ie - 1.0) * DXS;
}

// ---- Sound & Pitch: To PointProcess (cc) -------------------------------------------------------------------
// Praat walks the voiced stretches one after the other; inside a stretch the pulses are found one by one (each
// search starts at the previous pulse), but the stretches only interact through `added_right` (the last pulse
// added while walking right), which merely vetoes left-going pulses of later stretches.  So: (1) one wave per
// clip lists the stretches, (2) one wave per stretch walks it and records its pulses with their veto margins,
// (3) one wave per clip applies the vetoes in order and writes the pulses in ascending time.
struct Stretch { int il, irr, off, pad; };      // frame range, first slot of the stretch in the per-clip scratch

__device__ __forceinline__ bool voiced_at(const double* f, int nF, double ceiling, int i) {
    return i >= 0 && i < nF && f[i] > 0.0 && f[i] < ceiling;
}

// 256 threads: all four waves scan the samples for the absolute peak (a single wave took 7 500 dependent-ish rounds over a
// 30 s clip: most of this kernel's 3.2 ms), wave 0 then lists the stretches.
__global__ __launch_bounds__(256) void pulse_stretches_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ pci,
                                                              const double* __restrict__ sel_freq, double pdt, double ceiling,
                                                              Stretch* __restrict__ st, int max_st, int* __restrict__ n_st,
                                                              double* __restrict__ abs_peak) {
    __shared__ float s_pk[4];
    const ClipInfo c = pci[blockIdx.x];
    const int lane = threadIdx.x & 63, nF = c.n_frames;
    {   // Vector_getAbsoluteExtremum of the whole sound (no mean subtraction, unlike the pitch analysis)
        const float* x = wav + c.sample_off;
        float gp = 0.0f;                                      // |x| of float samples: exact in float
        const int n4 = c.n_samples >> 2;
        const bool al = (reinterpret_cast<uintptr_t>(x) & 15) == 0;   // 16-byte loads when the clip's first sample is 16-byte aligned
                                                                      // (the address itself: `wav` may be any float*, e.g. a sliced view)
        if (al) {
            const float4* x4 = reinterpret_cast<const float4*>(x);
            for (int i = threadIdx.x; i < n4; i += 256) {
                const float4 v = x4[i];
                gp = fmaxf(fmaxf(gp, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
            for (int i = 4 * n4 + threadIdx.x; i < c.n_samples; i += 256) gp = fmaxf(gp, fabsf(x[i]));
        } else {
            for (int i = threadIdx.x; i < c.n_samples; i += 256) gp = fmaxf(gp, fabsf(x[i]));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) gp = fmaxf(gp, __shfl_xor(gp, o, 64));
        if (lane == 0) s_pk[threadIdx.x >> 6] = gp;
        __syncthreads();
        if (threadIdx.x == 0) abs_peak[blockIdx.x] = (double)fmaxf(fmaxf(s_pk[0], s_pk[1]), fmaxf(s_pk[2], s_pk[3]));
        if (threadIdx.x >= 64) return;
    }
    const double* f = sel_freq + c.frame_off;
    Stretch* S = st + (int64_t)blockIdx.x * max_st;
    int count = 0;
    for (int base = 0; base < nF; base += 64) {
        const int i = base + lane;
        const bool v = voiced_at(f, nF, ceiling, i);
        const bool start = v && !voiced_at(f, nF, ceiling, i - 1), end = v && !voiced_at(f, nF, ceiling, i + 1);
        const unsigned long long ms = __ballot(start);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (start) { const int k = count + __popcll(ms & below); if (k < max_st) S[k].il = i; }
        if (end) {
            // the stretch that ends here is the last one started at or before this frame
            const int k = count + __popcll(ms & (below | (1ull << lane))) - 1;
            if (k >= 0 && k < max_st) S[k].irr = i;
        }
        count += __popcll(ms);
    }
    count = count < max_st ? count : max_st;
    __threadfence_block();
    // scratch slots: a stretch of n frames holds at most n*pdt*ceiling/0.8 + 3 pulses on either side
    int run = 0;
    for (int base = 0; base < count; base += 64) {
        const int k = base + lane;
        int cap = 0;
        if (k < count) cap = (int)((double)(S[k].irr - S[k].il + 1) * pdt * ceiling * 1.25) + 4;
        int inc = cap;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(inc, o, 64); if (lane >= o) inc += t2; }
        if (k < count) { S[k].off = run + inc - cap; S[k].pad = cap; }
        run += __shfl(inc, 63, 64);
    }
    if (lane == 0) n_st[blockIdx.x] = count;
}

// scratch per clip: left[slot] = (time, veto margin 0.8/f0) in walking order (entry 0 = the middle pulse),
// right[slot] = time; counts[stretch] = (n_left, n_right)
__global__ __launch_bounds__(256) void pulse_walk_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ pci,
                                                         const double* __restrict__ sel_freq, double pdt, double ceiling,
                                                         const double* __restrict__ abs_peak, const Stretch* __restrict__ st,
                                                         int max_st, const int* __restrict__ n_st, double2* __restrict__ left,
                                                         double* __restrict__ right, int cap_slots, int2* __restrict__ counts) {
    __shared__ float s_ps[4][PULSE_LDS];
    __shared__ double s_fw[4][PULSE_FW];                       // sliding window of the pitch track (the walk reads it once per pulse)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int clip = blockIdx.y, k = blockIdx.x * 4 + wv;
    if (k >= n_st[clip]) return;
    float* pwin = s_ps[wv];
    int64_t win0 = INT64_MIN;
    const ClipInfo c = pci[clip];
    const float* x = wav + c.sample_off;
    const int n = c.n_samples, nF = c.n_frames;
    const double* f = sel_freq + c.frame_off;
    const Stretch S = st[(int64_t)clip * max_st + k];
    double2* L = left + (int64_t)clip * cap_slots + S.off;
    double* R = right + (int64_t)clip * cap_slots + S.off;
    const int cap = S.pad;
    double* fw = s_fw[wv];
    int fw0 = -(1 << 30);                                     // frame index of fw[0]; far away = empty
    // Pitch "Get value at time" (pitch_value_at) on the LDS window: the two frames around t, refilled when the walk leaves it
    auto f0_at = [&](double t, int dir) -> double {
        const double qn = __longlong_as_double(0x7ff8000000000000LL);
        if (nF <= 0) return qn;
        const double ireal = (t - c.t1) / pdt;
        const int64_t ileft = (int64_t)floor(ireal);
        double phase = ireal - (double)ileft;
        int64_t inear, ifar;
        if (phase < 0.5) { inear = ileft; ifar = ileft + 1; } else { inear = ileft + 1; ifar = ileft; phase = 1.0 - phase; }
        if (inear < 0 || inear >= nF) return qn;
        const int64_t lo = ileft, hi = ileft + 1;              // both frames (either may lie outside the track: read as 0)
        if (lo < fw0 || hi >= fw0 + PULSE_FW) {
            const int64_t w0 = dir < 0 ? hi - (PULSE_FW - 1) : lo;
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < PULSE_FW; i += 64) { const int64_t j = w0 + i; fw[i] = (j >= 0 && j < nF) ? f[j] : 0.0; }
            fw0 = (int)w0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const double fn = fw[inear - fw0];
        if (!(fn > 0.0 && fn < ceiling)) return qn;
        if (ifar < 0 || ifar >= nF) return fn;
        const double ff = fw[ifar - fw0];
        if (!(ff > 0.0 && ff < ceiling)) return fn;
        return fn + phase * (ff - fn);
    };
    const double duration = c.xmax;                          // Pitch_getVoicedIntervalAfter works on the Pitch's domain = the sound's
    const double gp = abs_peak[clip];
    int nl = 0, nr = 0;
    double tleft = c.t1 + S.il * pdt - 0.5 * pdt, tright = c.t1 + S.irr * pdt + 0.5 * pdt;
    bool skip = tleft >= duration - 0.5 * pdt;               // Praat stops here; every later stretch starts even later
    tleft = tleft < 0.0 ? 0.0 : tleft;
    tright = tright > duration ? duration : tright;
    const double tmid = 0.5 * (tleft + tright);
    const double f0mid = pitch_value_at(f, nF, c.t1, pdt, ceiling, tmid);
    if (!(f0mid == f0mid)) skip = true;
    if (!skip) {
        double tmax = find_extremum_wave(x, n, c.x1, tmid - 0.5 / f0mid, tmid + 0.5 / f0mid, lane);
        if (lane == 0) L[0] = make_double2(tmax, 0.0);
        nl = 1;
        const double tsave = tmax;
        for (int g2 = 0; g2 < 200000; ++g2) {                      // to the left
            const double f0 = f0_at(tmax, -1);
            if (!(f0 == f0)) break;
            double peak, tout;
            const double corr = max_correlation_wave(x, n, c.x1, tmax, 1.0 / f0, tmax - 1.25 / f0, tmax - 0.8 / f0, lane, &tout, &peak, pwin, &win0, -1);
            tmax = tout;
            if (corr == -1.0) tmax -= 1.0 / f0;
            if (tmax < tleft) {
                if (corr > 0.7 && peak > 0.023333 * gp && nl < cap) { if (lane == 0) L[nl] = make_double2(tmax, 0.8 / f0); ++nl; }
                break;
            }
            if (corr > 0.3 && (peak == 0.0 || peak > 0.01 * gp) && nl < cap) { if (lane == 0) L[nl] = make_double2(tmax, 0.8 / f0); ++nl; }
        }
        tmax = tsave;
        for (int g2 = 0; g2 < 200000; ++g2) {                      // to the right
            const double f0 = f0_at(tmax, +1);
            if (!(f0 == f0)) break;
            double peak, tout;
            const double corr = max_correlation_wave(x, n, c.x1, tmax, 1.0 / f0, tmax + 0.8 / f0, tmax + 1.25 / f0, lane, &tout, &peak, pwin, &win0, +1);
            tmax = tout;
            if (corr == -1.0) tmax += 1.0 / f0;
            if (tmax > tright) {
                if (corr > 0.7 && peak > 0.023333 * gp && nr < cap) { if (lane == 0) R[nr] = tmax; ++nr; }
                break;
            }
            if (corr > 0.3 && (peak == 0.0 || peak > 0.01 * gp) && nr < cap) { if (lane == 0) R[nr] = tmax; ++nr; }
        }
    }
    if (lane == 0) counts[(int64_t)clip * max_st + k] = make_int2(nl, nr);
}

__global__ __launch_bounds__(64) void pulse_merge_kernel(const Stretch* __restrict__ st, int max_st, const int* __restrict__ n_st,
                                                         const double2* __restrict__ left, const double* __restrict__ right,
                                                         int cap_slots, const int2* __restrict__ counts,
                                                         double* __restrict__ pulses, int max_pulses, int* __restrict__ n_pulses) {
    const int clip = blockIdx.x, lane = threadIdx.x;
    const int ns = n_st[clip];
    double* pts = pulses + (int64_t)clip * max_pulses;
    int np_ = 0;
    double added_right = -1e308;
    for (int k = 0; k < ns; ++k) {
        const Stretch S = st[(int64_t)clip * max_st + k];
        const int2 cn = counts[(int64_t)clip * max_st + k];
        const double2* L = left + (int64_t)clip * cap_slots + S.off;
        const double* R = right + (int64_t)clip * cap_slots + S.off;
        if (cn.x <= 0) continue;
        // left-going pulses in ascending time = walking order reversed; entry 0 (the middle pulse) is never vetoed
        for (int base = cn.x - 1; base >= 1; base -= 64) {
            const int i = base - lane;
            bool keep = false;
            double t = 0.0;
            if (i >= 1) { const double2 e = L[i]; t = e.x; keep = t - added_right > e.y; }
            const unsigned long long m = __ballot(keep);
            const int pos = np_ + __popcll(m & ((1ull << lane) - 1ull));
            if (keep && pos < max_pulses) pts[pos] = t;
            np_ += __popcll(m);
        }
        if (np_ < max_pulses && lane == 0) pts[np_] = L[0].x;
        ++np_;
        for (int base = 0; base < cn.y; base += 64) {
            const int i = base + lane;
            if (i < cn.y && np_ + i < max_pulses) pts[np_ + i] = R[i];
        }
        if (cn.y > 0) added_right = R[cn.y - 1];
        np_ += cn.y;
    }
    if (lane == 0) n_pulses[clip] = np_ < max_pulses ? np_ : max_pulses;
}

// ---- Ltas (pitch-corrected) -> "Get slope" and robust tilt (src/mshds_extractor.py:227-251) ---------------
// One workgroup per clip; every wave takes every fourth pulse.  A pulse whose two neighbouring intervals are
// plausible periods contributes the energy spectrum of the one period around it: a DFT of exactly that many
// samples (lane = frequency bin, rotation recurrence over the samples), binned into 100 Hz bands.
// Per-wave band sums are combined in a fixed order, so the result does not depend on scheduling.
constexpr int LTAS_NB = 50;            // maximum frequency 5000 Hz / bandwidth 100 Hz
constexpr double LTAS_BW = 100.0;
constexpr int LTAS_MAXN = 1024;        // samples of one period that fit the LDS staging (longest period 20 ms = 320)

__device__ double ltas_mean_rect(const double* z, int nx, double x1, double dx, double xmin, double xmax) {
    const double qn = __longlong_as_double(0x7ff8000000000000LL);
    xmin = fmax(xmin, x1 - 0.5 * dx);
    xmax = fmin(xmax, x1 + (nx - 0.5) * dx);
    if (!(xmin < xmax)) return qn;
    const double rimin = (xmin - x1) / dx + 1.0, rimax = (xmax - x1) / dx + 1.0;
    double total = 0.0, rng = 0.0;
    if (rimax >= 0.5 && rimin < nx + 0.5) {
        const int imin = rimin < 0.5 ? 0 : (int)floor(rimin + 0.5);
        const int imax = rimax >= nx + 0.5 ? nx + 1 : (int)floor(rimax + 0.5);
        for (int i = imin + 1; i < imax; ++i) { rng += 1.0; total += z[i - 1]; }
        if (imin == imax) {
            if (imin >= 1 && imin <= nx) { const double ph = rimax - rimin; rng += ph; total += ph * z[imin - 1]; }
        } else {
            if (imin >= 1) { const double ph = imin - rimin + 0.5; rng += ph; total += ph * z[imin - 1]; }
            if (imax <= nx) { const double ph = rimax - imax + 0.5; rng += ph; total += ph * z[imax - 1]; }
        }
    }
    return rng > 0.0 ? total / rng : qn;
}

__global__ __launch_bounds__(256) void ltas_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                   const double* __restrict__ pulses, int max_pulses,
                                                   const int* __restrict__ n_pulses, double shortest, double longest,
                                                   double max_factor, double* __restrict__ out) {
    __shared__ double s_energy[4][LTAS_NB], s_count[4][LTAS_NB], s_z[LTAS_NB], s_slopes[LTAS_NB];
    __shared__ float s_x[4][LTAS_MAXN];
    __shared__ int s_periods[4], s_fail[4];
    const ClipInfo c = ci[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* x = wav + c.sample_off;
    const int n = c.n_samples;
    const double* pts = pulses + (int64_t)blockIdx.x * max_pulses;
    const int np_ = n_pulses[blockIdx.x];
    const double qn = __longlong_as_double(0x7ff8000000000000LL);
    for (int b = lane; b < LTAS_NB; b += 64) { s_energy[wv][b] = 0.0; s_count[wv][b] = 0.0; }
    int periods = 0, fail = 0;
    for (int ip = 1 + wv; ip < np_ - 1; ip += 4) {
        const double tl = pts[ip - 1], tm = pts[ip], tr = pts[ip + 1];
        const double left = tm - tl, right = tr - tm;
        const double factor = left > right ? left / right : right / left;
        if (!(left >= shortest && left <= longest && right >= shortest && right <= longest && factor <= max_factor)) continue;
        const double t1 = tm - 0.5 * left, t2 = tm + 0.5 * right;
        const int64_t ix1 = (int64_t)ceil((t1 - c.x1) / DXS), ix2 = (int64_t)floor((t2 - c.x1) / DXS);   // Sound_extractPart
        if (ix2 < ix1 || ix2 - ix1 + 1 > LTAS_MAXN) { fail = 1; continue; }   // Praat: "no samples" aborts the analysis
        const int m = (int)(ix2 - ix1 + 1);
        for (int j = lane; j < m; j += 64) { const int64_t i = ix1 + j; s_x[wv][j] = (i >= 0 && i < n) ? x[i] : 0.0f; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const double sdx = 1.0 / (DXS * m);
        const int nfreq = m / 2 + 1;
        // bins k = 1 .. nfreq-1 whose band ceil(k*sdx/100) is within 1..50 (k = 0 falls into band 0)
        for (int kb = 1; kb < nfreq; kb += 64) {
            const int k = kb + lane;
            const double freq = k * sdx;
            int band = (int)ceil(freq / LTAS_BW);
            const bool on = k < nfreq && band >= 1 && band <= LTAS_NB;
            double e = 0.0;
            if (__any(on)) {
                // sum_j x_j exp(-2 pi i k j / m): rotate (c, s) by the bin's angle, which lies in (0, pi]
                const double th = 2.0 * PI * (double)(k < nfreq ? k : 0) / (double)m;
                const double C = cos_0_pi(th), S = sin_0_pi(th);
                double cr = 1.0, sr = 0.0, re = 0.0, im = 0.0;
                for (int j = 0; j < m; ++j) {
                    const double v = s_x[wv][j];
                    re += v * cr; im -= v * sr;
                    const double c2 = cr * C - sr * S;
                    sr = sr * C + cr * S;
                    cr = c2;
                }
                re *= DXS; im *= DXS;
                e = (re * re + im * im) * 2.0 * sdx;
            }
            if (!on) { band = -1 - lane; e = 0.0; }
            // bands are non-decreasing in k: the first lane of a run adds the whole run (<= 4 bins per band)
            const int bprev = __shfl_up(band, 1, 64);
            const bool head = on && (lane == 0 || bprev != band);
            double sum = e, cnt = 1.0;
#pragma unroll
            for (int d = 1; d <= 7; ++d) {
                const int bn = __shfl_down(band, d, 64);
                const double en = __shfl_down(e, d, 64);
                if (lane + d < 64 && bn == band) { sum += en; cnt += 1.0; }
            }
            // a run can continue in the next 64-bin round: LDS accumulation below handles that (same wave, ordered)
            if (head) { s_energy[wv][band - 1] += sum; s_count[wv][band - 1] += cnt; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        ++periods;
    }
    if (lane == 0) { s_periods[wv] = periods; s_fail[wv] = fail; }
    __syncthreads();
    if (tid == 0) {
        const int total_periods = s_periods[0] + s_periods[1] + s_periods[2] + s_periods[3];
        const int failed = s_fail[0] | s_fail[1] | s_fail[2] | s_fail[3];
        double slope = qn, tilt = qn;
        if (np_ - 2 >= 1 && total_periods >= 1 && !failed) {
            double total = 0.0;
            for (int b = 0; b < LTAS_NB; ++b) {
                s_energy[0][b] = (s_energy[0][b] + s_energy[1][b]) + (s_energy[2][b] + s_energy[3][b]);
                s_count[0][b] = (s_count[0][b] + s_count[1][b]) + (s_count[2][b] + s_count[3][b]);
                total += s_count[0][b];
            }
            const double duration = c.xmax;                      // PointProcess_Sound_to_Ltas divides by sound->xmax - sound->xmin
            bool any = false;
            for (int b = 0; b < LTAS_NB; ++b) {
                if (s_count[0][b] > 0.0) {
                    const double mean_e = s_energy[0][b] / s_count[0][b];
                    s_z[b] = 10.0 * log10(mean_e * (total / LTAS_NB) / LTAS_BW / duration / 4.0e-10);
                    any = true;
                } else {
                    s_z[b] = qn;
                }
            }
            if (any) {
                for (int b = 0; b < LTAS_NB; ++b) s_slopes[b] = s_z[b];     // defined values before filling
                for (int b = 0; b < LTAS_NB; ++b) {
                    if (s_slopes[b] == s_slopes[b]) continue;
                    int bl = b - 1, br = b + 1;
                    while (bl >= 0 && !(s_slopes[bl] == s_slopes[bl])) --bl;
                    while (br < LTAS_NB && !(s_slopes[br] == s_slopes[br])) ++br;
                    if (bl < 0) s_z[b] = s_slopes[br];
                    else if (br >= LTAS_NB) s_z[b] = s_slopes[bl];
                    else s_z[b] = ((br - b) * s_slopes[bl] + (b - bl) * s_slopes[br]) / (double)(br - bl);
                }
                const double x1 = 0.5 * LTAS_BW;
                const double low = ltas_mean_rect(s_z, LTAS_NB, x1, LTAS_BW, 50.0, 1000.0);
                const double high = ltas_mean_rect(s_z, LTAS_NB, x1, LTAS_BW, 1000.0, 4000.0);
                slope = high - low;
                // Theil's incomplete method over the bands centred in [100, 5000] Hz
                int imin = 1 + (int)ceil((100.0 - x1) / LTAS_BW), imax = 1 + (int)floor((5000.0 - x1) / LTAS_BW);
                imin = imin < 1 ? 1 : imin;
                imax = imax > LTAS_NB ? LTAS_NB : imax;
                const int cntp = imax - imin + 1, nc = cntp / 2, n2 = (cntp & 1) ? nc + 1 : nc;
                for (int i = 0; i < nc; ++i) {
                    const double xa = x1 + (imin - 1 + i) * LTAS_BW, xb = x1 + (imin - 1 + n2 + i) * LTAS_BW;
                    s_slopes[i] = (s_z[imin - 1 + n2 + i] - s_z[imin - 1 + i]) / (xb - xa);
                }
                for (int i = 1; i < nc; ++i) {                              // insertion sort (<= 24 values)
                    const double v = s_slopes[i];
                    int j = i - 1;
                    while (j >= 0 && s_slopes[j] > v) { s_slopes[j + 1] = s_slopes[j]; --j; }
                    s_slopes[j + 1] = v;
                }
                if (nc >= 1) {                                              // NUMquantile(0.5)
                    if (nc == 1) tilt = s_slopes[0];
                    else {
                        const double place = 0.5 * nc + 0.5;
                        int lf = (int)floor(place);
                        lf = lf < 1 ? 1 : (lf > nc - 1 ? nc - 1 : lf);
                        tilt = s_slopes[lf] == s_slopes[lf - 1] ? s_slopes[lf - 1]
                                                                  : s_slopes[lf - 1] + (place - lf) * (s_slopes[lf] - s_slopes[lf - 1]);
                    }
                } else {
                    slope = qn;                                             // the tilt report fails -> both NaN (:250-251)
                }
            }
        }
        out[2 * blockIdx.x] = slope;
        out[2 * blockIdx.x + 1] = tilt;
    }
}

// ---- _measureFormants statistics: F1, B1, F2, B2 linearly interpolated at every pulse ------------------------
__global__ __launch_bounds__(64) void formant_stats_kernel(const FormantFrame* __restrict__ ff, const ClipInfo* __restrict__ fci,
                                                           double fdt, const double* __restrict__ pulses, int max_pulses,
                                                           const int* __restrict__ n_pulses, double* __restrict__ out) {
    const ClipInfo c = fci[blockIdx.x];
    const int lane = threadIdx.x, np_ = n_pulses[blockIdx.x], nF = c.n_frames;
    const FormantFrame* F = ff + c.frame_off;
    const double* pts = pulses + (int64_t)blockIdx.x * max_pulses;
    const double qn = __longlong_as_double(0x7ff8000000000000LL);
    double cnt[4] = {0, 0, 0, 0}, sum[4] = {0, 0, 0, 0};
    auto value = [&](int k, double t) -> double {      // k: 0 F1, 1 B1, 2 F2, 3 B2
        if (nF <= 0) return qn;
        const double ireal = (t - c.t1) / fdt;
        const int64_t ileft = (int64_t)floor(ireal);
        double phase = ireal - (double)ileft;
        int64_t inear, ifar;
        if (phase < 0.5) { inear = ileft; ifar = ileft + 1; } else { inear = ileft + 1; ifar = ileft; phase = 1.0 - phase; }
        if (inear < 0 || inear >= nF) return qn;
        const int fi = k >> 1;
        const double vn = (k & 1) ? F[inear].b[fi] : F[inear].f[fi];
        if (!(vn == vn)) return qn;
        if (ifar < 0 || ifar >= nF) return vn;
        const double vf = (k & 1) ? F[ifar].b[fi] : F[ifar].f[fi];
        if (!(vf == vf)) return vn;
        return vn + phase * (vf - vn);
    };
    for (int i = lane; i < np_; i += 64)
        for (int k = 0; k < 4; ++k) { const double v = value(k, pts[i]); if (v == v) { cnt[k] += 1; sum[k] += v; } }
    double mean[4];
    for (int k = 0; k < 4; ++k) { cnt[k] = wave_sum_f64(cnt[k]); sum[k] = wave_sum_f64(sum[k]); mean[k] = cnt[k] > 0 ? sum[k] / cnt[k] : qn; }
    double sq[4] = {0, 0, 0, 0};
    for (int i = lane; i < np_; i += 64)
        for (int k = 0; k < 4; ++k) { const double v = value(k, pts[i]); if (v == v) { const double d = v - mean[k]; sq[k] += d * d; } }
    for (int k = 0; k < 4; ++k) sq[k] = wave_sum_f64(sq[k]);
    if (lane == 0)
        for (int k = 0; k < 4; ++k) {
            out[blockIdx.x * 8 + 2 * k] = mean[k];
            out[blockIdx.x * 8 + 2 * k + 1] = cnt[k] > 1 ? sqrt(sq[k] / (cnt[k] - 1)) : qn;
        }
}

}  // namespace mshds
}  // namespace rsaf

using namespace rsaf;
using namespace rsaf::mshds;

extern "C" {

int rsaf_mshds_frameout_doubles(void) { return (int)(sizeof(FrameOut) / sizeof(double)); }

int rsaf_mshds_clip_peak(const float* wav, const void* clip_info, int n_clips, double* gpeak, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0, "negative n_clips");
    if (n_clips == 0) return RSAF_OK;
    RSAF_CHECK_ARG(wav && clip_info && gpeak, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("mshds_clip_peak", s, 0.0, 0.0);
    hipLaunchKernelGGL(clip_peak_kernel, dim3(n_clips), dim3(256), 0, s, wav, (const ClipInfo*)clip_info, gpeak);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int rsaf_mshds_intensity(const float* wav, const void* clip_info, int n_clips, int max_frames, const double* window,
                         int half_window, double time_step, int subtract_mean, double* db_out, double* stats_out,
                         rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && n_clips <= 65535 && max_frames >= 0, "bad clip/frame count");
    if (n_clips == 0) return RSAF_OK;
    RSAF_CHECK_ARG(wav && clip_info && window && db_out && stats_out, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (max_frames > 0) {
        ProfScope prof("mshds_intensity", s, 0.0, 0.0);
        hipLaunchKernelGGL(intensity_kernel, dim3((max_frames + 3) / 4, n_clips), dim3(256), 0, s, wav,
                           (const ClipInfo*)clip_info, window, half_window, time_step, subtract_mean, db_out);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(intensity_stats_kernel, dim3(n_clips), dim3(64), 0, s, db_out, (const ClipInfo*)clip_info, stats_out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

// W_N^k = exp(-2 pi i k / N), k < N / 2, in double precision (host libm), one table per (device, N), kept for the
// life of the process (pitch_ac_kernel)
static int fft_twiddles(int N, const double** out) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, double*> cache;
    int dev = 0;
    RSAF_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({dev, N});
    if (it == cache.end()) {
        std::vector<double> h((size_t)N);                          // N / 2 complex numbers
        for (int k = 0; k < N / 2; ++k) {
            const double a = 2.0 * M_PI * (double)k / (double)N;
            h[2 * k] = cos(a);
            h[2 * k + 1] = -sin(a);
        }
        if (N >= 4) { h[2 * (N / 4)] = 0.0; h[2 * (N / 4) + 1] = -1.0; }      // exactly -i
        double* d = nullptr;
        RSAF_CHECK_HIP(hipMalloc(&d, (size_t)N * sizeof(double)));
        RSAF_CHECK_HIP(hipMemcpy(d, h.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice));
        it = cache.emplace(std::make_pair(dev, N), d).first;
    }
    *out = it->second;
    return RSAF_OK;
}

// second_*: optional outputs of the same analysis with another voicing threshold (h2_voicing_thr >= 0): the frame
// kernel shares the correlation and the refinement, the path finder runs once per threshold
// workspace per frame: the correlation row, the coefficient blocks of two candidate lists, the frame record
static inline int64_t pitch_ws_bytes_per_frame(int rstride) {
    return (int64_t)rstride * (int64_t)sizeof(double) + 2 * (int64_t)PC_DOUBLES * (int64_t)sizeof(double) + (int64_t)HDR_INTS * (int64_t)sizeof(int);
}

static int pitch_impl(const float* wav, const void* clip_info, int n_clips, int max_frames, const double* gpeak,
                      const double* window, const double* window_r, const double* params_host, void* frame_out,
                      unsigned char* psi, int* end_state, double* sel_freq, double* sel_strength, double* stats_out,
                      double voicing_thr2, void* frame_out2, unsigned char* psi2, int* end_state2, double* sel_freq2,
                      double* sel_strength2, double* stats_out2, const double* sinc_cheb, void* workspace,
                      int64_t workspace_bytes, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && n_clips <= 65535 && max_frames >= 0, "bad clip/frame count");
    if (n_clips == 0) return RSAF_OK;
    RSAF_CHECK_ARG(wav && clip_info && gpeak && params_host && frame_out && psi && end_state && sel_freq &&
                   sel_strength && stats_out, "NULL pointer");
    const bool dual = voicing_thr2 >= 0.0;
    RSAF_CHECK_ARG(!dual || (frame_out2 && psi2 && end_state2 && sel_freq2 && sel_strength2 && stats_out2),
                   "NULL pointer (second threshold outputs)");
    const double* h = params_host;
    PitchParams P;
    P.dt = h[0]; P.min_pitch = h[1]; P.ceiling = h[2]; P.voicing_thr = h[3]; P.octave_cost = h[4];
    const double silence_thr = h[5], octave_jump = h[6], vuv = h[7];
    P.nsamp_window = (int)h[8]; P.nsamp_period = (int)h[9]; P.min_lag = (int)h[10]; P.max_lag = (int)h[11];
    P.brent_ixmax = (int)h[12]; P.max_cand = (int)h[13]; P.refine_depth = (int)h[14]; P.is_cc = (int)h[15];
    P.dt_window = h[16];
    const int table_mode = (int)h[17];            // 0: shared table only, 1: + the tables of the clipped depths, 2: one table per cell
    P.cheb_clipped = table_mode == 1 ? 1 : 0;
    P.voicing_thr2 = dual ? voicing_thr2 : -1.0;
    { const char* e = getenv("RSAF_PITCH_STOP"); P.debug_stop = e ? atoi(e) : 0; }
    P.refine_margin = 0.0;   // lazy refinement is off: it changed a few frames' selection (parity first)
    P.half_window = P.nsamp_window / 2;
    P.half_period = P.nsamp_period / 2 + 1;
    RSAF_CHECK_ARG(P.max_cand >= 2 && P.max_cand <= MAXC - 1, "max_candidates must be in [2, 15]");
    RSAF_CHECK_ARG(P.nsamp_window >= 4 && P.brent_ixmax >= 2 && P.max_lag >= 2, "window too short");
    RSAF_CHECK_ARG(P.is_cc || (window && window_r), "AC needs the window tables");
    RSAF_CHECK_ARG((P.is_cc ? P.max_lag : P.brent_ixmax) <= 1023, "more than 1023 lags (pitch floor below ~16 Hz) is not supported");
    P.nfft = 1;                                                    // Praat: while (nsampFFT < nsamp_window * (1 + 0.5)) nsampFFT *= 2
    while ((double)P.nfft < (double)P.nsamp_window * 1.5) P.nfft *= 2;
    if (P.nfft < 16) P.nfft = 16;
    RSAF_CHECK_ARG(P.is_cc || P.nsamp_window + P.brent_ixmax <= P.nfft, "brent_ixmax must not exceed half the analysis window");
    const int seg_len = P.is_cc ? P.nsamp_window + P.max_lag + 1 : P.nsamp_window;
    const int Lr = P.is_cc ? P.max_lag : P.brent_ixmax;
    const int rstride = Lr + 2;                                    // r[0..L] + the frame's relative intensity
    int ncc = 64;                                                  // CC: complex FFT length, >= nw + max_lag + 1
    while (ncc < seg_len) ncc *= 2;
    RSAF_CHECK_ARG(!P.is_cc || ncc <= 4096, "cross-correlation window + lag range longer than 4 095 samples is not supported");
    RSAF_CHECK_ARG(P.is_cc || P.nfft <= 4096, "autocorrelation window longer than 2 730 samples is not supported");
    const size_t lds_corr = P.is_cc ? (size_t)ncc * 2 * 2 * sizeof(double) + (size_t)(((Lr + 2) & ~1) + 32) * sizeof(double)   // two complex buffers + sumy2 + scratch
                                    : (size_t)(2 * P.nfft + 8) * sizeof(double);
    // in-kernel refinement (the form before the refinement kernels existed): the A/B reference of the tests
    const bool defer = getenv("RSAF_PITCH_INKERNEL") == nullptr;
    const bool grouped = defer && table_mode == 2 && !dual && P.is_cc && sinc_cheb != nullptr;
    int r_lo_h, r_hi_h;                                            // (as in the kernel: the depth-30 estimates' reach in grouped mode)
    pitch_r_range(P.brent_ixmax, Lr, P.min_lag, P.max_lag, grouped ? 30 : P.refine_depth, &r_lo_h, &r_hi_h);
    const size_t lds_cand = (size_t)(((r_hi_h - r_lo_h + 2) & ~1) + 3 * MAX_MAXIMA + 6 * MAXC + MAXC * 2 * NCH) * sizeof(double) +
                            (size_t)(MAX_MAXIMA + 2 * MAXC + 4) * sizeof(int);
    RSAF_CHECK_ARG(lds_corr <= 150 * 1024 && lds_cand <= 150 * 1024, "analysis window too long for LDS");
    // the correlation rows of a group of clips live in the caller's workspace between the kernels, and behind them what the
    // candidate kernel leaves for the refinement kernels: per frame the coefficient blocks of two lists and a 128-byte record
    const int64_t row_bytes_per_clip = (int64_t)max_frames * pitch_ws_bytes_per_frame(rstride);
    RSAF_CHECK_ARG(max_frames == 0 || (workspace && workspace_bytes >= row_bytes_per_clip),
                   "workspace too small (rsaf_mshds_pitch_workspace_bytes)");
    int group = max_frames == 0 ? n_clips : (int)std::min<int64_t>(n_clips, workspace_bytes / std::max<int64_t>(row_bytes_per_clip, 1));
    if (max_frames > 0) group = (int)std::min<int64_t>(group, ((int64_t)1 << 26) / max_frames > 0 ? ((int64_t)1 << 26) / max_frames : 1);   // frame index in 27 bits (cell queue)
    RSAF_CHECK_ARG(table_mode != 2 || (!dual && P.is_cc), "per-cell tables serve single-threshold cross-correlation analyses only");
    hipStream_t s = (hipStream_t)stream;
    int log2m = 0;
    while ((2 << log2m) < P.nfft) ++log2m;                           // nfft = 2 M = 2^(log2m + 1)
    int log2n = 0;
    while ((1 << log2n) < ncc) ++log2n;
    // transform lengths of 512 .. 2048 complex points run one wave per frame (RSAF_PITCH_FFT=wg: the workgroup kernels)
    int wave_r = 0;
    {
        const char* e = getenv("RSAF_PITCH_FFT");
        const bool want = !(e && e[0] == 'w' && e[1] == 'g');
#ifndef RSAF_TEST_KERNELS
        // the workgroup-FFT kernels the one-wave kernels superseded (transforms of up to 2 048 points) are compiled only into
        // a test build (RSAF_BUILD_TEST_KERNELS=1 python -m ...build), where they serve as an independent A/B check
        RSAF_CHECK_ARG(want, "RSAF_PITCH_FFT=wg needs a library built with RSAF_BUILD_TEST_KERNELS=1");
#endif
        // A shorter transform is zero-padded up to the smallest wave size: the correlation is linear as long as the lags stay
        // below (transform length - window), so a longer transform returns the same values (autocorrelation: 512 complex =
        // 1024 real points; cross-correlation: 1024 points, whose transform back has the 512 the wave kernel needs).
        if (want && !P.is_cc && log2m <= 11) wave_r = log2m <= 9 ? 8 : 1 << (log2m - 6);
        if (want && P.is_cc && log2n <= 11) wave_r = log2n <= 10 ? 16 : 32;
    }
    if (wave_r && !P.is_cc) P.nfft = 128 * wave_r;                  // 2 S real points
    if (wave_r && P.is_cc) ncc = 64 * wave_r;
    if (!wave_r && lds_corr > 48 * 1024) {
#ifdef RSAF_TEST_KERNELS
        const void* fn = (const void*)pitch_ac_kernel<11>;            // 4 096 points: 64 KB (the only AC instance above 48 KB)
        if (P.is_cc) fn = log2n == 11 ? (const void*)pitch_cc_kernel<11> : (const void*)pitch_cc_kernel<12>;   // 64 / 128 KB
#else
        const void* fn = (const void*)pitch_cc_kernel<12>;            // 4 096-point cross-correlation: the one transform above the wave sizes
#endif
        RSAF_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_corr));
    }
    const double* twiddles = nullptr;
    const double* twiddles2 = nullptr;
    if (!P.is_cc) {
        const int rc = fft_twiddles(P.nfft, &twiddles);
        if (rc != RSAF_OK) return rc;
    } else {
        int rc = fft_twiddles(2 * ncc, &twiddles);                  // W_2N^k: the N-point complex transform
        if (rc != RSAF_OK) return rc;
        rc = fft_twiddles(ncc, &twiddles2);                         // W_N^k: the N/2-point transform and the spectrum pass
        if (rc != RSAF_OK) return rc;
    }
    if (lds_cand > 48 * 1024)
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)pitch_cand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_cand));
    // the Chebyshev form needs the full depth on both sides of every cell a candidate can use
    const double* cheb = table_mode == 2 ? nullptr : sinc_cheb;
    const int cell_lag_lo = P.min_lag > 2 ? P.min_lag : 2;
    const int cell_lag_hi = std::min(P.max_lag - 1, P.brent_ixmax - 1);
    {
        const int lag_lo = cell_lag_lo, lag_hi = cell_lag_hi;
        const bool unclipped = P.brent_ixmax + lag_lo - 1 >= P.refine_depth && lag_hi + 2 + P.refine_depth <= P.brent_ixmax;
        P.cheb_all_full = unclipped ? 1 : 0;
        if (cheb == nullptr) P.cheb_clipped = 0;
        // clipped analyses keep the Chebyshev form only with the per-depth tables behind the shared one
        if ((!unclipped && !P.cheb_clipped) || getenv("RSAF_PITCH_NO_CHEB")) cheb = nullptr;
    }
    // per-cell tables (mshds.sinc_cell_tables): cells b_lo .. b_hi, the lags 0 .. L (r is symmetric: the two taps that meet
    // a lag are summed in the table) padded to a multiple of four
    const int cell_b_lo = P.brent_ixmax + cell_lag_lo - 1, cell_n_b = cell_lag_hi - cell_lag_lo + 2;
    const int cell_ntap_pad = (Lr + 1 + 3) & ~3;
    const size_t lds_cell = (size_t)cell_ntap_pad * NCH * sizeof(double) + 4 * CELL_Q * sizeof(unsigned);
    if (grouped) {
        RSAF_CHECK_ARG(cell_n_b >= 1 && lds_cell <= 150 * 1024, "per-cell tables: lag range too long for LDS");
        if (lds_cell > 48 * 1024)
            RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)pitch_cell_coef_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds_cell));
    }
    if (max_frames > 0) {
        // algorithmic flops of the correlation kernels, counted for equal-length clips (an upper bound for ragged batches);
        // the candidate kernel's work is not counted
        // AC: two complex FFTs of M = nfft / 2 points (5 M log2 M flops each) and the spectrum pass (~30 flops per point)
        const double Mfft = 0.5 * (double)P.nfft;
        const double ac_flops = 2.0 * 5.0 * Mfft * log2(Mfft) + 30.0 * Mfft;
        // CC: one complex FFT of ncc points, one of ncc / 2, the spectrum pass (~40 flops per point) and the prefix sums
        const double cc_flops = 5.0 * ncc * log2((double)ncc) + 2.5 * ncc * log2(0.5 * ncc) + 40.0 * 0.5 * ncc + 4.0 * seg_len;
        // LDS bytes a frame moves through the FFT kernel (every pass reads and writes its N complex doubles: log4 stages of
        // each transform, the staging pass and the spectrum pass): the kernel's own roofline is the LDS, not the FLOPs
        // (one wave per frame, wave_fft.h: two exchanges per transform, each writing and reading the S complex doubles, and the
        // paired spectrum step: 160 S bytes per autocorrelation frame, 120 S + the running sums per cross-correlation frame)
        const double ac_lds = wave_r ? 160.0 * Mfft : 16.0 * Mfft * (2.0 * ceil(log2(Mfft) / 2.0) + 3.0) * 2.0;
        const double cc_lds = wave_r ? 120.0 * ncc + 16.0 * Lr
                                     : 16.0 * ncc * (ceil(log2((double)ncc) / 2.0) + 2.0) * 2.0 + 16.0 * 0.5 * ncc * (ceil(log2(0.5 * ncc) / 2.0) + 1.0) * 2.0;
        for (int c0 = 0; c0 < n_clips; c0 += group) {
            const int nc = std::min(group, n_clips - c0);
            const ClipInfo* cig = (const ClipInfo*)clip_info + c0;
            {
            // family "mshds_pitch_{ac,cc}_fft": the correlation kernel alone (FLOPs = its FFTs, bytes = its LDS traffic)
            ProfScope prof(P.is_cc ? "mshds_pitch_cc_fft" : "mshds_pitch_ac_fft", s,
                           (P.is_cc ? cc_flops : ac_flops) * (double)max_frames * (double)nc,
                           (P.is_cc ? cc_lds : ac_lds) * (double)max_frames * (double)nc);
            if (wave_r) {
                // one wave per frame (wave_fft.h): S = 64 R complex points
                const dim3 grid((max_frames + WF_FRAMES - 1) / WF_FRAMES, nc);
                const double2_t* twz = reinterpret_cast<const double2_t*>(P.is_cc ? twiddles2 : twiddles);
                if (P.is_cc) {
                    const size_t lds_w = (size_t)((wave_r == 16 ? wfft::Plan<16>::LDS_DOUBLES : wfft::Plan<32>::LDS_DOUBLES) + ((Lr + 3) & ~1)) * sizeof(double);
                    if (wave_r == 16)
                        hipLaunchKernelGGL(pitch_cc_wave_kernel<16>, grid, dim3(64), lds_w, s, wav, cig, gpeak + c0, P, twz,
                                           (double*)workspace, rstride, max_frames);
                    else
                        hipLaunchKernelGGL(pitch_cc_wave_kernel<32>, grid, dim3(64), lds_w, s, wav, cig, gpeak + c0, P, twz,
                                           (double*)workspace, rstride, max_frames);
                } else {
#define RSAF_ACW_CASE(RR)                                                                                             \
    case RR:                                                                                                          \
        hipLaunchKernelGGL(pitch_ac_wave_kernel<RR>, grid, dim3(64), (size_t)wfft::Plan<RR>::LDS_DOUBLES * sizeof(double), s, wav, \
                           cig, gpeak + c0, window, window_r, P, twz, (double*)workspace, rstride, max_frames);        \
        break;
                    switch (wave_r) { RSAF_ACW_CASE(8) RSAF_ACW_CASE(16) RSAF_ACW_CASE(32) default: break; }
#undef RSAF_ACW_CASE
                }
            } else if (P.is_cc) {
#define RSAF_CC_CASE(LG)                                                                                              \
    case LG:                                                                                                          \
        hipLaunchKernelGGL(pitch_cc_kernel<LG>, dim3((max_frames + CC_FRAMES_PER_WG - 1) / CC_FRAMES_PER_WG, nc),      \
                           dim3(256), lds_corr, s, wav, cig, gpeak + c0, P,                                           \
                           reinterpret_cast<const double2_t*>(twiddles), reinterpret_cast<const double2_t*>(twiddles2), \
                           (double*)workspace, rstride, max_frames);                                                  \
        break;
                switch (log2n) {
#ifdef RSAF_TEST_KERNELS
                    RSAF_CC_CASE(6) RSAF_CC_CASE(7) RSAF_CC_CASE(8) RSAF_CC_CASE(9) RSAF_CC_CASE(10) RSAF_CC_CASE(11)
#endif
                    RSAF_CC_CASE(12)
                    default: set_error("rsaf_mshds_pitch: unsupported FFT length"); return RSAF_ERR_ARG;
                }
#undef RSAF_CC_CASE
            } else {
#define RSAF_AC_CASE(LG)                                                                                              \
    case LG:                                                                                                          \
        hipLaunchKernelGGL(pitch_ac_kernel<LG>, dim3((max_frames + AC_FRAMES_PER_WG - 1) / AC_FRAMES_PER_WG, nc),      \
                           dim3(256), lds_corr, s, wav, cig, gpeak + c0,                                              \
                           window, window_r, P, reinterpret_cast<const double2_t*>(twiddles), (double*)workspace,     \
                           rstride, max_frames);                                                                      \
        break;
                switch (log2m) {
#ifdef RSAF_TEST_KERNELS
                    RSAF_AC_CASE(3) RSAF_AC_CASE(4) RSAF_AC_CASE(5) RSAF_AC_CASE(6) RSAF_AC_CASE(7) RSAF_AC_CASE(8)
                    RSAF_AC_CASE(9) RSAF_AC_CASE(10) RSAF_AC_CASE(11)
#endif
                    default: set_error("rsaf_mshds_pitch: unsupported FFT length"); return RSAF_ERR_ARG;
                }
#undef RSAF_AC_CASE
            }
            RSAF_CHECK_HIP(hipGetLastError());
            }
            // family "mshds_pitch_cand": maxima, candidate lists, Brent refinement.  Work model: every frame's normalised
            // correlation row comes back from the HBM workspace ((Lr + 2) doubles); on the Chebyshev path the coefficient build
            // of the frame's candidates runs on the fp64 matrix pipe: ceil(2 depth / 4) tap groups x 2 column tiles of
            // v_mfma_f64_16x16x4_f64 (2 048 flops each).  The Brent iterations themselves (a dozen polynomial evaluations per
            // candidate) and the direct path's sinc sums are not counted.
            const double cand_rows = (double)max_frames * (double)nc;
            // workspace of the group: rows | coefficients of list A | of list B | frame records
            const int64_t gframes = (int64_t)nc * max_frames;
            double* pc_a = (double*)workspace + gframes * rstride;
            double* pc_b = pc_a + gframes * PC_DOUBLES;
            int* hdr = reinterpret_cast<int*>(pc_b + gframes * PC_DOUBLES);
            DeferArgs DA{defer ? hdr : nullptr, defer ? pc_a : nullptr, defer ? pc_b : nullptr, grouped ? 1 : 0};
            {
            ProfScope prof(grouped ? "mshds_pitch_cand_lists" : cheb ? "mshds_pitch_cand_cheb" : "mshds_pitch_cand_direct", s,
                           cheb ? cand_rows * 2048.0 * 2.0 * ceil(2.0 * P.refine_depth / 4.0) : 0.0,
                           cand_rows * (double)(Lr + 2) * 8.0);
            hipLaunchKernelGGL(pitch_cand_kernel, dim3(max_frames, nc), dim3(CT), lds_cand, s, cig, gpeak + c0, P,
                               (const double*)workspace, rstride, max_frames, (FrameOut*)frame_out,
                               dual ? (FrameOut*)frame_out2 : (FrameOut*)nullptr, cheb, DA);
            RSAF_CHECK_HIP(hipGetLastError());
            }
            if (defer && P.debug_stop == 0) {
                if (grouped) {
                    // one table per cell on the fp64 matrix pipe: <= 28 cells per frame x (2 L + 1) taps x 16 coefficients
                    static const int chunk_frames = [] { const char* e = getenv("RSAF_PITCH_CELL_CHUNK"); const int v = e ? atoi(e) : 4096; return v >= 256 && v % 256 == 0 ? v : 4096; }();
                    const int64_t n_chunks = (gframes + chunk_frames - 1) / chunk_frames;
                    const int64_t n_wg = 8 * (int64_t)cell_n_b * ((n_chunks + 7) / 8);
                    RSAF_CHECK_ARG(n_wg <= 0x7fffffffLL, "per-cell tables: too many workgroups");
                    ProfScope prof("mshds_pitch_cand_cells", s, cand_rows * 28.0 * 2.0 * cell_ntap_pad * NCH,   // <= 28 cells per frame
                                   cand_rows * 28.0 * (double)(Lr + 1) * 8.0);
                    hipLaunchKernelGGL(pitch_cell_coef_kernel, dim3((unsigned)n_wg), dim3(256), lds_cell, s, cig, nc, max_frames,
                                       (const int*)hdr, (const double*)workspace, rstride, Lr, P.brent_ixmax, sinc_cheb,
                                       cell_b_lo, cell_n_b, cell_ntap_pad, chunk_frames, (int)n_chunks, pc_a);
                    RSAF_CHECK_HIP(hipGetLastError());
                }
                const bool low_is_second = dual && P.voicing_thr2 < P.voicing_thr;
                FrameOut* oa = low_is_second ? (FrameOut*)frame_out2 : (FrameOut*)frame_out;
                FrameOut* ob = !dual ? (FrameOut*)nullptr : low_is_second ? (FrameOut*)frame_out : (FrameOut*)frame_out2;
                ProfScope prof("mshds_pitch_cand_brent", s, 0.0, cand_rows * 15.0 * 2.0 * NCH * 8.0);
                hipLaunchKernelGGL(pitch_brent_kernel, dim3((max_frames + BR_FRAMES - 1) / BR_FRAMES, nc), dim3(64), 0, s, cig, (const int*)hdr,
                                   (const double*)pc_a, (const double*)pc_b, max_frames, P.brent_ixmax, oa, ob);
                RSAF_CHECK_HIP(hipGetLastError());
            }
        }
    }
    for (int pass = 0; pass < (dual ? 2 : 1); ++pass) {
        const FrameOut* fo = (const FrameOut*)(pass ? frame_out2 : frame_out);
        unsigned char* ps = pass ? psi2 : psi;
        int* es = pass ? end_state2 : end_state;
        double* sf = pass ? sel_freq2 : sel_freq;
        double* ss = pass ? sel_strength2 : sel_strength;
        const double vt = pass ? voicing_thr2 : P.voicing_thr;
        {
            ProfScope prof("mshds_pitch_path", s, 0.0, 0.0);
            hipLaunchKernelGGL(path_kernel, dim3(n_clips), dim3(64), 0, s, fo, (const ClipInfo*)clip_info, P.dt, silence_thr,
                               vt, P.octave_cost, octave_jump, vuv, P.ceiling, ps, es);
            RSAF_CHECK_HIP(hipGetLastError());
            hipLaunchKernelGGL(backtrack_kernel, dim3(n_clips), dim3(256), 0, s, fo, (const ClipInfo*)clip_info, ps, es, sf, ss);
            RSAF_CHECK_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(pitch_stats_kernel, dim3(n_clips), dim3(64), 0, s, sf, (const ClipInfo*)clip_info, P.ceiling,
                           pass ? stats_out2 : stats_out);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    return RSAF_OK;
}

// bytes of correlation rows per clip (max_frames rows of max_lag + 2 doubles); the analysis runs the clips in groups of
// floor(workspace_bytes / this), so any multiple >= 1 works and n_clips multiples avoid the grouping
int64_t rsaf_mshds_pitch_workspace_bytes_per_clip(int max_frames, const double* params_host /* 18 doubles */) {
    if (!params_host || max_frames < 0) return -1;
    const int Lr = (int)params_host[15] ? (int)params_host[11] : (int)params_host[12];
    return (int64_t)max_frames * pitch_ws_bytes_per_frame(Lr + 2);
}

int rsaf_mshds_pitch(const float* wav, const void* clip_info, int n_clips, int max_frames, const double* gpeak,
                     const double* window, const double* window_r, const double* params_host /* 18 doubles */,
                     void* frame_out, unsigned char* psi, int* end_state, double* sel_freq, double* sel_strength, double* stats_out,
                     const double* sinc_cheb, void* workspace, int64_t workspace_bytes, rsaf_stream_t stream) {
    return pitch_impl(wav, clip_info, n_clips, max_frames, gpeak, window, window_r, params_host, frame_out, psi, end_state,
                      sel_freq, sel_strength, stats_out, -1.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sinc_cheb,
                      workspace, workspace_bytes, stream);
}

int rsaf_mshds_pitch_dual(const float* wav, const void* clip_info, int n_clips, int max_frames, const double* gpeak,
                          const double* window, const double* window_r, const double* params_host /* 18 doubles */,
                          void* frame_out, unsigned char* psi, int* end_state, double* sel_freq, double* sel_strength,
                          double* stats_out, double voicing_threshold2, void* frame_out2, unsigned char* psi2, int* end_state2,
                          double* sel_freq2, double* sel_strength2, double* stats_out2, const double* sinc_cheb,
                          void* workspace, int64_t workspace_bytes, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(voicing_threshold2 >= 0.0, "second voicing threshold must be >= 0");
    return pitch_impl(wav, clip_info, n_clips, max_frames, gpeak, window, window_r, params_host, frame_out, psi, end_state,
                      sel_freq, sel_strength, stats_out, voicing_threshold2, frame_out2, psi2, end_state2, sel_freq2,
                      sel_strength2, stats_out2, sinc_cheb, workspace, workspace_bytes, stream);
}

// peaks of an n-frame contour: at most n/2; the LDS form keeps SR_MAX_PEAKS of them (a smooth 16 ms contour of at most
// ~8 500 frames has far fewer), the global-memory form of long clips sizes the lists exactly
static int sr_peak_cap(int max_frames, bool in_global) { return in_global ? max_frames / 2 + 2 : SR_MAX_PEAKS; }
static bool sr_in_global(int max_frames) {
    return (size_t)2 * ((max_frames + 1) & ~1) * sizeof(double) + SR_MAX_PEAKS * sizeof(int) > 150 * 1024;
}

int64_t rsaf_mshds_speechrate_workspace_doubles(int max_frames) {
    const bool g = sr_in_global(max_frames);
    const int64_t cap = sr_peak_cap(max_frames, g);
    return 3 * ((int64_t)max_frames + 2) + 2 * cap + (g ? 2 * ((int64_t)max_frames + 2) + cap / 2 + 2 : 0);
}

int rsaf_mshds_speechrate(const double* intensity_db, const void* clip_info, int n_clips, int max_frames,
                          double intensity_dt, const double* sel_freq, const void* pitch_clip_info, double pitch_dt,
                          double pitch_ceiling, double* workspace, double* out, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && max_frames >= 0, "bad clip/frame count");
    if (n_clips == 0) return RSAF_OK;
    RSAF_CHECK_ARG(intensity_db && clip_info && sel_freq && pitch_clip_info && workspace && out, "NULL pointer");
    const bool in_global = sr_in_global(max_frames);
    const int peak_cap = sr_peak_cap(max_frames, in_global);
    const size_t lds = in_global ? 0 : (size_t)2 * ((max_frames + 1) & ~1) * sizeof(double) + SR_MAX_PEAKS * sizeof(int);
    hipStream_t s = (hipStream_t)stream;
    if (lds > 48 * 1024)
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)speechrate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds));
    ProfScope prof("mshds_speechrate", s, 0.0, 0.0);
    hipLaunchKernelGGL(speechrate_kernel, dim3(n_clips), dim3(64), lds, s, intensity_db, (const ClipInfo*)clip_info,
                       intensity_dt, sel_freq, (const ClipInfo*)pitch_clip_info, pitch_dt, pitch_ceiling, workspace,
                       rsaf_mshds_speechrate_workspace_doubles(max_frames), max_frames, peak_cap, in_global ? 1 : 0, out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int rsaf_mshds_resample10k_table_stride(int depth) { return (2 * depth + 1 + 8 * (RS_QL - 1) + 7) / 8 * 8 + 8; }

int rsaf_mshds_resample10k(const double* lowpassed, const void* resample_info, int n_clips, int max_out, const double* tables,
                           int table_stride, const int* phase_base, int depth, double* out, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && n_clips <= 65535 && max_out >= 0 && depth >= 3, "bad argument");
    if (n_clips == 0 || max_out == 0) return RSAF_OK;
    RSAF_CHECK_ARG(lowpassed && resample_info && tables && phase_base && out, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const int taps = 2 * depth + 1;
    RSAF_CHECK_ARG(table_stride >= rsaf_mshds_resample10k_table_stride(depth), "weight rows shorter than rsaf_mshds_resample10k_table_stride");
    const int span = 8 * (RS_QT - 1) + 8 + taps + 8 * RS_QL + 8;       // phase bases differ by < 8; the tap loop runs past `taps`
    const size_t lds = (size_t)(span + span / 32 + 2) * sizeof(double);
    RSAF_CHECK_ARG(lds <= 150 * 1024, "resampler depth too large for LDS");
    if (lds > 48 * 1024)
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)resample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope prof("mshds_resample10k", s, 0.0, 0.0);
    const int nq = (max_out + 4) / 5;
    hipLaunchKernelGGL(resample_kernel, dim3((nq + RS_QT - 1) / RS_QT, n_clips), dim3(320), lds, s, lowpassed,
                       (const ResampleInfo*)resample_info, tables, table_stride, phase_base, depth, out);
    RSAF_CHECK_HIP(hipGetLastError());
    const int n_edge = (int)((double)(depth + 2) * 0.625) + 3;         // output samples within depth + 2 input samples of an end
    hipLaunchKernelGGL(resample_edge_kernel, dim3((2 * n_edge + 255) / 256, n_clips), dim3(256), 0, s, lowpassed,
                       (const ResampleInfo*)resample_info, depth, n_edge, 1.6, out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int rsaf_mshds_formants(const double* y10, const void* resample_info, const void* clip_info, int n_clips, int max_frames,
                        const double* window, int nsamp_window, double time_step, double dx_out, double preemph_factor,
                        void* frames_out, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && n_clips <= 65535 && max_frames >= 0 && nsamp_window >= 16, "bad argument");
    if (n_clips == 0 || max_frames == 0) return RSAF_OK;
    RSAF_CHECK_ARG(y10 && resample_info && clip_info && window && frames_out, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)4 * 2 * (nsamp_window + 2) * sizeof(double);
    RSAF_CHECK_ARG(lds <= 150 * 1024, "formant window too long");
    if (lds > 48 * 1024)
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)formant_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope prof("mshds_formant_frames", s, 0.0, 0.0);
    hipLaunchKernelGGL(formant_kernel, dim3((max_frames + 4 * FB_GROUP - 1) / (4 * FB_GROUP), n_clips), dim3(256), lds, s, y10,
                       (const ResampleInfo*)resample_info, (const ClipInfo*)clip_info, window, nsamp_window, time_step,
                       dx_out, preemph_factor, (FormantFrame*)frames_out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

// scratch of rsaf_mshds_pulses: stretch table + per-stretch counts + left (time, margin) + right (time) slots
static void pulses_layout(int n_clips, int max_frames, int max_samples, double pitch_dt, double ceiling, int* max_st,
                          int* cap_slots, int64_t* total) {
    *max_st = max_frames / 2 + 2;
    *cap_slots = (int)((double)max_frames * pitch_dt * ceiling * 1.25) + 4 * *max_st + 16;
    (void)max_samples;
    *total = (int64_t)n_clips * ((int64_t)*max_st * (sizeof(Stretch) + sizeof(int2)) + sizeof(int) * 2 + sizeof(double) +
                                 (int64_t)*cap_slots * (sizeof(double2) + sizeof(double))) + 256;
}

int64_t rsaf_mshds_pulses_workspace_bytes(int n_clips, int max_frames, double pitch_dt, double pitch_ceiling) {
    int ms, cs;
    int64_t total;
    pulses_layout(n_clips, max_frames, 0, pitch_dt, pitch_ceiling, &ms, &cs, &total);
    return total;
}

int rsaf_mshds_pulses(const float* wav, const void* pitch_clip_info, int n_clips, int max_frames, const double* sel_freq,
                      double pitch_dt, double pitch_ceiling, void* workspace, int64_t workspace_bytes, double* pulses,
                      int max_pulses, int* n_pulses, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && n_clips <= 65535 && max_pulses >= 1 && max_frames >= 0, "bad argument");
    if (n_clips == 0) return RSAF_OK;
    RSAF_CHECK_ARG(wav && pitch_clip_info && sel_freq && workspace && pulses && n_pulses, "NULL pointer");
    int max_st, cap_slots;
    int64_t need;
    pulses_layout(n_clips, max_frames, 0, pitch_dt, pitch_ceiling, &max_st, &cap_slots, &need);
    RSAF_CHECK_ARG(workspace_bytes >= need, "workspace too small (rsaf_mshds_pulses_workspace_bytes)");
    char* w = (char*)workspace;
    double2* left = (double2*)w;              w += (int64_t)n_clips * cap_slots * sizeof(double2);
    double* right = (double*)w;               w += (int64_t)n_clips * cap_slots * sizeof(double);
    double* abs_peak = (double*)w;            w += (int64_t)n_clips * sizeof(double);
    Stretch* st = (Stretch*)w;                w += (int64_t)n_clips * max_st * sizeof(Stretch);
    int2* counts = (int2*)w;                  w += (int64_t)n_clips * max_st * sizeof(int2);
    int* n_st = (int*)w;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("mshds_pulses", s, 0.0, 0.0);
    hipLaunchKernelGGL(pulse_stretches_kernel, dim3(n_clips), dim3(256), 0, s, wav, (const ClipInfo*)pitch_clip_info, sel_freq,
                       pitch_dt, pitch_ceiling, st, max_st, n_st, abs_peak);
    RSAF_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(pulse_walk_kernel, dim3((max_st + 3) / 4, n_clips), dim3(256), 0, s, wav, (const ClipInfo*)pitch_clip_info,
                       sel_freq, pitch_dt, pitch_ceiling, abs_peak, st, max_st, n_st, left, right, cap_slots, counts);
    RSAF_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(pulse_merge_kernel, dim3(n_clips), dim3(64), 0, s, st, max_st, n_st, left, right, cap_slots, counts, pulses,
                       max_pulses, n_pulses);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int rsaf_mshds_ltas_slope_tilt(const float* wav, const void* clip_info, int n_clips, const double* pulses,
                               int max_pulses, const int* n_pulses, double shortest_period, double longest_period,
                               double max_period_factor, double* out, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && max_pulses >= 0, "bad clip/pulse count");
    if (n_clips == 0) return RSAF_OK;
    RSAF_CHECK_ARG(wav && clip_info && pulses && n_pulses && out, "NULL pointer");
    RSAF_CHECK_ARG(longest_period * 16000.0 + 2.0 <= LTAS_MAXN, "longest period does not fit the LDS staging");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("mshds_ltas", s, 0.0, 0.0);
    hipLaunchKernelGGL(ltas_kernel, dim3(n_clips), dim3(256), 0, s, wav, (const ClipInfo*)clip_info, pulses, max_pulses,
                       n_pulses, shortest_period, longest_period, max_period_factor, out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int rsaf_mshds_formant_stats(const void* frames, const void* clip_info, int n_clips, double time_step, const double* pulses,
                             int max_pulses, const int* n_pulses, double* out, rsaf_stream_t stream) {
    if (n_clips <= 0) return RSAF_OK;
    RSAF_CHECK_ARG(frames && clip_info && pulses && n_pulses && out, "NULL pointer");
    hipLaunchKernelGGL(formant_stats_kernel, dim3(n_clips), dim3(64), 0, (hipStream_t)stream, (const FormantFrame*)frames,
                       (const ClipInfo*)clip_info, time_step, pulses, max_pulses, n_pulses, out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int rsaf_mshds_hnr_mean(const double* sel_freq, const double* sel_strength, const void* clip_info, int n_clips,
                        double* out, rsaf_stream_t stream) {
    if (n_clips <= 0) return RSAF_OK;
    RSAF_CHECK_ARG(sel_freq && sel_strength && clip_info && out, "NULL pointer");
    hipLaunchKernelGGL(hnr_stats_kernel, dim3(n_clips), dim3(64), 0, (hipStream_t)stream, sel_freq, sel_strength,
                       (const ClipInfo*)clip_info, out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int rsaf_mshds_spectral_moments(const float* wav, const void* clip_info, const void* pitch_clip_info, int n_clips,
                                int max_frames, const double* sel_freq, double pitch_dt, double ceiling,
                                const double* window, const double* twiddle, int nsamp_window, int nfft, int nbins,
                                double time_step, double freq_step, double* moments, double* stats_out,
                                rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && n_clips <= 65535 && max_frames >= 0, "bad clip/frame count");
    if (n_clips == 0) return RSAF_OK;
    RSAF_CHECK_ARG(wav && clip_info && pitch_clip_info && sel_freq && window && twiddle && moments && stats_out,
                   "NULL pointer");
    RSAF_CHECK_ARG(nfft > 0 && (nfft & (nfft - 1)) == 0 && nbins > 0 && nbins <= nfft / 2 + 1, "bad FFT geometry");
    hipStream_t s = (hipStream_t)stream;
    RSAF_CHECK_ARG(nfft >= 2 && (nfft & (nfft - 1)) == 0 && nfft >= nsamp_window && nbins <= nfft / 2 + 1, "nfft must be a power of two >= window");
    const size_t lds = (size_t)(nfft + nbins) * sizeof(double);
    RSAF_CHECK_ARG(lds <= 60 * 1024, "spectrogram window too long");
    if (max_frames > 0) {
        ProfScope prof("mshds_spec_moments", s, 0.0, 0.0);
        const int threads = nfft >= 2048 ? 512 : 256;      // nfft / 4 butterflies per pass of the half-length FFT
        const char* e = getenv("RSAF_SPM_WAVE");
        if (nfft == 1024 && nsamp_window >= 2 && !(e && e[0] == '0'))
            hipLaunchKernelGGL(spec_moments_wave_kernel, dim3((max_frames + SPM_FRAMES - 1) / SPM_FRAMES, n_clips), dim3(64), 0, s,
                               wav, (const ClipInfo*)clip_info, (const ClipInfo*)pitch_clip_info, sel_freq, pitch_dt, ceiling,
                               window, (const double2*)twiddle, nsamp_window, nsamp_window / 2, nbins, time_step, freq_step,
                               moments);
        else
        hipLaunchKernelGGL(spec_moments_kernel, dim3(max_frames, n_clips), dim3(threads), lds, s, wav,
                           (const ClipInfo*)clip_info, (const ClipInfo*)pitch_clip_info, sel_freq, pitch_dt, ceiling,
                           window, (const double2*)twiddle, nsamp_window, nsamp_window / 2, nfft, nbins, time_step,
                           freq_step, moments);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(moments_stats_kernel, dim3(n_clips), dim3(64), 0, s, moments, (const ClipInfo*)clip_info, stats_out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

}  // extern "C"
