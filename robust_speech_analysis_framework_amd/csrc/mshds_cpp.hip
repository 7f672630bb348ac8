// Cepstral peak prominence of the voiced stretches (MSHDS a8) for gfx950.
//
// Replaces, for a packed batch of clips resident in HBM, the Praat calls of src/mshds_extractor.py:270-297:
//   PointProcess "To TextGrid (vuv)" 0.02 0.1 -> "Down to Table" (6 decimals) -> Sound.extract_part ->
//   "To PowerCepstrogram" 60 0.002 5000 50 -> "Get CPPS" no 0.01 0.001 60 330 0.05 parabolic 0.001 0 Straight Robust
//   -> mean of the per-interval values above 4 dB.
// Kernels: segment table (one wave per clip), Praat's FFT low-pass of every interval (three passes, praat_lowpass.h),
// sinc interpolation to 10 kHz (one thread per output sample),
// power cepstrum per frame (two 1024-point fp64 FFTs in LDS), smoothed CPP per frame (moving averages,
// two bitonic sorts for Theil's line, parabolic peak), reduction per interval and clip.
// The arithmetic is fp64 and nothing is contracted to FMA where Praat's rounding decides an integer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "praat_interp.h"
#include "praat_lowpass.h"
#include "rsaf_common.h"
#include "wave_fft.h"

#pragma clang fp contract(off)

namespace rsaf {
namespace mshds_cpp {

constexpr double DXS = 1.0 / 16000.0;
constexpr double PI = 3.14159265358979323846;
constexpr double FS_OUT = 10000.0;
constexpr double DXO = 1.0 / FS_OUT;
constexpr int DEPTH = 50;                 // Sound_resample precision of Sound_to_PowerCepstrogram
constexpr double DT = 0.002;              // cepstrogram time step
constexpr double DQ = 1.0e-4;             // quefrency step = 1 / FS_OUT
constexpr int NFFT_MAX = 1024;            // 0.1 s window at 10 kHz = 1000 samples
constexpr int NQ_MAX = NFFT_MAX / 2 + 1;  // 513
constexpr int SEG_DOUBLES = 15;

struct ClipInfo {      // same 48-byte rows as csrc/mshds.hip
    int64_t sample_off;
    int64_t frame_off;
    double t1;
    int n_samples;
    int n_frames;
    double x1;         // time of the first sample
    double xmax;       // end of the sound's time domain
};

// one voiced interval (all fields double so that the table is one plain array)
struct Seg {
    double ix1, m_in, m_out, res_off, frame_off, nf, x1_seg, x1o, window, t1, nx, nfft;
    double work_off, item_off, lg;   // low-pass transform: first complex number, first work item, log2 of its length
};
static_assert(sizeof(Seg) == SEG_DOUBLES * sizeof(double), "Seg layout");

// "%.6f" and back ("Down to Table" prints the interval times with 6 decimals, the reference parses them again).
// printf rounds the exact binary value; interval ends clipped to the sound's end are multiples of 62.5 us, i.e.
// sit within one ulp of a decimal tie, so the product t * 1e6 must not be rounded before the decision: the FMA
// recovers its rounding error exactly.
__device__ __forceinline__ double round6(double t) {
    const double p = t * 1.0e6;
    const double e = fma(t, 1.0e6, -p);                  // t * 1e6 == p + e exactly
    double q = floor(p);
    double f = (p - q) + e;                              // fractional part of the exact product
    if (f < 0.0) { q -= 1.0; f += 1.0; }
    const bool up = f > 0.5 || (f == 0.5 && fmod(q, 2.0) != 0.0);
    return (up ? q + 1.0 : q) / 1.0e6;
}

// work items (workgroups) of one pass of an interval's low-pass transform of 2^lg samples: the larger of the column and row pass
__host__ __device__ inline int lp_items(int lg) {
    const resample::LpGeom g = resample::lp_geom(lg);
    const int cols = (1 << g.log2) / g.C, rows = (1 << g.log1) / 2 + 1;
    return cols > rows ? cols : rows;
}

// ---- 1. voiced intervals -> segment table -----------------------------------------------------------------
// hdr[clip] = {n_seg, fail, total_frames, total_resampled}
__global__ __launch_bounds__(64) void segments_kernel(const ClipInfo* __restrict__ ci, const double* __restrict__ pulses,
                                                      int max_pulses, const int* __restrict__ n_pulses, double max_period,
                                                      double mean_period, double pitch_floor, Seg* __restrict__ segs,
                                                      int max_seg, int cap_res, int cap_frames, int64_t cap_work, int lg_max,
                                                      int* __restrict__ hdr) {
    const ClipInfo c = ci[blockIdx.x];
    const int lane = threadIdx.x;
    const double* t = pulses + (int64_t)blockIdx.x * max_pulses;
    const int np_ = n_pulses[blockIdx.x];
    Seg* out = segs + (int64_t)blockIdx.x * max_seg;
    const double xmax = c.xmax, half = 0.5 * mean_period;         // PointProcess_to_TextGrid_vuv: the sound's domain
    int nseg = 0, fail = 0;
    int64_t res_off = 0, frame_off = 0, work_off = 0, item_off = 0;
    double begin_voiceless = 0.0;
    int cur_start = 0;
    // every lane keeps the same (uniform) state; lane 0 writes
    auto emit = [&](int first, int last) {
        double end_voiceless = t[first] - half;
        if (end_voiceless <= begin_voiceless) end_voiceless = begin_voiceless;
        const double begin_voiced = end_voiceless;
        double end_voiced = t[last] + half;
        if (end_voiced > xmax) end_voiced = xmax;
        begin_voiceless = end_voiced;
        const double tmin = round6(begin_voiced), tmax = round6(end_voiced);
        if (tmin >= tmax) return;                                            // :284
        const int64_t ix1 = (int64_t)ceil((tmin - c.x1) / DXS), ix2 = (int64_t)floor((tmax - c.x1) / DXS);
        const double dur = tmax - tmin;
        const int64_t m_out = (int64_t)floor(dur * FS_OUT + 0.5);
        if (ix2 < ix1 || m_out < 1) { fail = 1; return; }                    // Praat raises outside the inner try -> NaN
        const int64_t m_in = ix2 - ix1 + 1;
        double window = 2.0 * 3.0 / pitch_floor;
        const double my_duration = DXS * (double)m_in;
        if (window > my_duration) window = my_duration;
        const int64_t nf = (int64_t)floor((my_duration - window) / DT) + 1;
        const double x1_seg = c.x1 + (double)ix1 * DXS - tmin;
        const double mid = x1_seg - 0.5 * DXS + 0.5 * my_duration;
        const double t1 = mid - 0.5 * (double)nf * DT + 0.5 * DT;
        const int64_t nx = (int64_t)floor(window * FS_OUT + 0.5);
        int nfft = 2;
        while (nfft < nx) nfft *= 2;
        int lg = 11;                                                         // Sound_resample: first power of two >= n + 2000
        while (((int64_t)1 << lg) < m_in + 2 * resample::ANTI_TURN_AROUND) ++lg;
        const int64_t work_len = (int64_t)1 << (lg - 1);
        if (nseg >= max_seg || res_off + m_out > cap_res || frame_off + nf > cap_frames || nx > NFFT_MAX || nx < 1 || nf < 1 ||
            lg > lg_max || work_off + work_len > cap_work) {
            fail = 1;
            return;
        }
        if (lane == 0) {
            Seg s;
            s.ix1 = (double)ix1; s.m_in = (double)m_in; s.m_out = (double)m_out; s.res_off = (double)res_off;
            s.frame_off = (double)frame_off; s.nf = (double)nf; s.x1_seg = x1_seg;
            s.x1o = 0.5 * (dur - (double)(m_out - 1) * DXO);
            s.window = window; s.t1 = t1; s.nx = (double)nx; s.nfft = (double)nfft;
            s.work_off = (double)work_off; s.item_off = (double)item_off; s.lg = (double)lg;
            out[nseg] = s;
        }
        ++nseg;
        res_off += m_out;
        frame_off += nf;
        work_off += work_len;
        item_off += lp_items(lg);
    };
    for (int base = 0; base < np_; base += 64) {
        const int i = base + lane;
        const bool brk = i >= 1 && i < np_ && (t[i] - t[i - 1] > max_period);
        unsigned long long m = __ballot(brk);
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            emit(cur_start, base + b - 1);
            cur_start = base + b;
        }
    }
    if (np_ > 0) emit(cur_start, np_ - 1);
    if (lane == 0) {
        hdr[4 * blockIdx.x + 0] = nseg;
        hdr[4 * blockIdx.x + 1] = fail;
        hdr[4 * blockIdx.x + 2] = (int)frame_off;
        hdr[4 * blockIdx.x + 3] = (int)res_off;
    }
}

// index of the segment that owns element g of a prefix-summed range (OFF = index of the offset field in Seg)
template <int OFF>
__device__ __forceinline__ int find_seg(const Seg* __restrict__ s, int nseg, int g) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)reinterpret_cast<const double*>(s + mid)[OFF] <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ---- 2. 16 kHz -> 10 kHz of every interval: Sound_resample (10000, 50) --------------------------------------
// FFT low-pass of the extracted part (its own transform length), then NUM_interpolate_sinc on the grid centred in the
// part's domain.  The intervals are found on the device, so the three transform passes run as loops over work items:
// item w of a clip = workgroup (w - item_off) of the interval that owns it; every pass uses the same item count.
// Those launches take the long intervals only (transforms above 2^14 samples); a shorter interval is low-passed by one
// workgroup inside LDS (seg_lowpass_lds_kernel).
template <int PASS>
__global__ __launch_bounds__(256) void seg_lowpass_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                          const Seg* __restrict__ segs, int max_seg, const int* __restrict__ hdr,
                                                          resample::c64* __restrict__ work, int64_t cap_work,
                                                          double* __restrict__ lowpassed, int64_t lp_origin,
                                                          int lg_lo, int lg_hi, resample::LpTables T) {
    extern __shared__ resample::c64 lp_lds[];
    const int clip = blockIdx.y;
    const int nseg = hdr[4 * clip];
    if (nseg <= 0) return;
    const Seg* S = segs + (int64_t)clip * max_seg;
    const ClipInfo c = ci[clip];
    const int total = (int)S[nseg - 1].item_off + lp_items((int)S[nseg - 1].lg);
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const Seg s = S[find_seg<13>(S, nseg, w)];
        if ((int)s.lg < lg_lo || (int)s.lg > lg_hi) continue;               // the other launch's class (same for the whole workgroup)
        resample::LpSig sg;
        sg.in_off = c.sample_off + (int64_t)s.ix1;
        sg.out_off = c.sample_off - lp_origin + (int64_t)s.ix1;
        sg.work_off = (int64_t)clip * cap_work + (int64_t)s.work_off;
        sg.n = (int)s.m_in;
        sg.lg = (int)s.lg;
        const int bx = w - (int)s.item_off;
        if (PASS == 0) resample::lp_cols_body<false>(wav, work, lowpassed, sg, bx, T, lp_lds);
        else if (PASS == 1) resample::lp_rows_body(work, sg, bx, FS_OUT * DXS, T, lp_lds);
        else resample::lp_cols_body<true>(wav, work, lowpassed, sg, bx, T, lp_lds);
        __syncthreads();                                                     // the next item reuses the LDS buffer
    }
}

// intervals whose whole transform fits LDS (up to 2^14 samples = 8 192 complex numbers = 128 KB): one workgroup per interval,
// samples in, low-passed samples out, nothing through the work buffer
__global__ __launch_bounds__(1024) void seg_lowpass_lds_kernel(const float* __restrict__ wav, const ClipInfo* __restrict__ ci,
                                                               const Seg* __restrict__ segs, int max_seg, const int* __restrict__ hdr,
                                                               double* __restrict__ lowpassed, int64_t lp_origin, int lg_hi,
                                                               resample::LpTables T) {
    extern __shared__ resample::c64 lp_lds[];
    const int clip = blockIdx.y;
    const int nseg = hdr[4 * clip];
    const Seg* S = segs + (int64_t)clip * max_seg;
    const ClipInfo c = ci[clip];
    for (int k = blockIdx.x; k < nseg; k += gridDim.x) {
        const Seg s = S[k];
        if ((int)s.lg > lg_hi) continue;                                     // a long interval: the three-pass launches
        resample::lp_whole_in_lds(wav + c.sample_off + (int64_t)s.ix1, lowpassed + (c.sample_off - lp_origin) + (int64_t)s.ix1,
                                  (int)s.m_in, (int)s.lg, FS_OUT * DXS, T, lp_lds);
    }
}

// 16 kHz -> 10 kHz is a step of 1.6 input samples per output: the fractional position repeats every 5 outputs (8 input samples),
// so an interval has five sets of 2 * DEPTH weights.  A workgroup whose 256 outputs lie in one interval builds the five sets
// once (500 weights, two per thread) and stages its 508-sample input window in LDS; an output with the full depth on both
// sides is then 100 multiply-adds.  Outputs near the ends of an interval (clipped depth), workgroups that straddle two
// intervals and phases whose fraction lies within 1e-6 of an integer take Praat's formula term by term as before
// (`RSAF_CPP_POLYPHASE=0`: everything does).  Against the term-by-term form the sums differ by rounding only (the weights
// are those of the phase's first output: positions 8 samples apart differ by ~1e-11 in the fraction).
constexpr int RS_TAPS = 2 * DEPTH;                    // 100
constexpr int RS_WIN = 512;                            // input window of a workgroup: 255 * 1.6 + 100 samples
__global__ __launch_bounds__(256) void resample_kernel(const double* __restrict__ lowpassed, int64_t lp_origin,
                                                       const ClipInfo* __restrict__ ci, const Seg* __restrict__ segs, int max_seg,
                                                       const int* __restrict__ hdr, int cap_res, double* __restrict__ res,
                                                       int polyphase) {
    __shared__ double s_w[RS_TAPS * 5];
    __shared__ double s_in[RS_WIN];
    __shared__ int s_si0;
    const int clip = blockIdx.y;
    const int nseg = hdr[4 * clip], total = hdr[4 * clip + 3];
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * 256;
    if (g0 >= total || nseg <= 0) return;               // workgroup-uniform
    const int g = g0 + tid;
    const bool live = g < total;
    const Seg* S = segs + (int64_t)clip * max_seg;
    const int si = find_seg<3>(S, nseg, live ? g : total - 1);
    const Seg s = S[si];
    const ClipInfo c = ci[clip];
    const double* y = lowpassed + (c.sample_off - lp_origin) + (int64_t)s.ix1;   // the extracted part, low-passed
    const int64_t n = (int64_t)s.m_in;
    const int i = (live ? g : total - 1) - (int)s.res_off;
    const double pos = (s.x1o + (double)i * DXO - s.x1_seg) / DXS;       // real 0-based index into the extracted part
    const double x = pos + 1.0;                                          // Praat's 1-based index
    double* out = res + (int64_t)clip * cap_res + g;
    if (tid == 0) s_si0 = si;
    __syncthreads();
    const bool one_seg = __syncthreads_and(si == s_si0) != 0;            // (threads past the clip's last output repeat it)
    if (!polyphase || !one_seg) {
        if (live) *out = praat_interpolate_sinc(y, n, x, DEPTH);
        return;
    }
    constexpr double PI_ = 3.14159265358979323846;
    const int64_t midleft = (int64_t)floor(x);
    const double frac = x - (double)midleft;
    const int i0 = g0 - (int)s.res_off;                                  // output index of thread 0 within the interval
    // fraction of the first output of the workgroup that has phase r = (i0 + q) mod 5, q < 5 (the same expression thread q evaluates)
    auto phase_frac = [&](int q) {
        const double xr = (s.x1o + (double)(i0 + q) * DXO - s.x1_seg) / DXS + 1.0;
        return xr - floor(xr);
    };
    for (int e = tid; e < RS_TAPS * 5; e += 256) {
        const int t = e / 5, r = e - 5 * t;                               // weight of tap t for phase r
        const double fr = phase_frac((r - i0 % 5 + 5) % 5);
        double wt = 0.0;
        if (fr > 1e-6 && fr < 1.0 - 1e-6) {
            // left half: tap t < DEPTH is k = DEPTH - 1 - t samples left of midleft; right half: k = t - DEPTH right of midright
            const bool left = t < DEPTH;
            const int k = left ? DEPTH - 1 - t : t - DEPTH;
            const double a0 = PI_ * (left ? fr : 1.0 - fr), span = (left ? fr : 1.0 - fr) + (double)DEPTH;
            const double a = a0 + PI_ * (double)k;
            double hs = 0.5 * sin(a0);
            if (k & 1) hs = -hs;
            wt = hs / a * (1.0 + cos(a / span));
        }
        s_w[e] = wt;
    }
    const double fr_mine = phase_frac(tid % 5);                          // the first output of this thread's phase
    const bool phase_ok = fr_mine > 1e-6 && fr_mine < 1.0 - 1e-6 && fabs(fr_mine - frac) < 1e-9;
    // input window of the workgroup: 0-based samples base ... base + RS_WIN - 1, base = midleft(thread 0) - DEPTH
    const double x0 = (s.x1o + (double)i0 * DXO - s.x1_seg) / DXS + 1.0;
    const int64_t base = (int64_t)floor(x0) - DEPTH;
    for (int e = tid; e < RS_WIN; e += 256) {
        const int64_t j = base + e;
        s_in[e] = (j >= 0 && j < n) ? y[j] : 0.0;
    }
    __syncthreads();
    if (!live) return;
    const int64_t midright = midleft + 1;
    const bool full = x <= (double)n && x >= 1.0 && midright - 1 >= DEPTH && n - midleft >= DEPTH;
    const int off = (int)(midleft - DEPTH - base);                       // window element of tap 0 (sample midleft - DEPTH + 1, 1-based)
    if (!full || !phase_ok || off < 0 || off + RS_TAPS > RS_WIN) {
        *out = praat_interpolate_sinc(y, n, x, DEPTH);
        return;
    }
    const int r = (i0 + tid) % 5;
    double acc = 0.0;
#pragma unroll 4
    for (int t = 0; t < RS_TAPS; ++t) acc += s_w[5 * t + r] * s_in[off + t];
    *out = acc;
}

// ---- 3. power cepstrum of every frame ----------------------------------------------------------------------
// in-place radix-2 FFT of n complex values that were stored in bit-reversed order (tw[k] = exp(-2 pi i k / 1024))
__device__ void fft_inplace(double2* a, int n, int log2n, const double2* __restrict__ tw, int tid) {
    for (int st = 1; st <= log2n; ++st) {
        const int half = 1 << (st - 1);
        const int tstep = NFFT_MAX >> st;
        for (int b = tid; b < (n >> 1); b += 256) {
            const int grp = b >> (st - 1), p = b & (half - 1);
            const int i0 = (grp << st) + p, i1 = i0 + half;
            const double2 w = tw[p * tstep];
            const double2 u = a[i0], v = a[i1];
            const double tr = v.x * w.x - v.y * w.y, ti = v.x * w.y + v.y * w.x;
            a[i0] = make_double2(u.x + tr, u.y + ti);
            a[i1] = make_double2(u.x - tr, u.y - ti);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ wfft::cplx ld_c(const double2* __restrict__ tw, int i) {
    const double2 w = tw[i];
    return wfft::cplx{w.x, w.y};
}

__device__ __forceinline__ int bitrev(int i, int log2n) { return log2n ? (int)(__brev((unsigned)i) >> (32 - log2n)) : 0; }

// Real-input FFT of n = 2m points from an m-point complex FFT of z[j] = x[2j] + i x[2j+1] (already transformed in `a`):
// X[k] = E + W_n^k O,  E = (Z[k] + conj Z[m-k]) / 2,  O = -i (Z[k] - conj Z[m-k]) / 2,  k = 0..m  (Z[m] = Z[0]).
__device__ __forceinline__ double2 real_fft_bin(const double2* a, int m, int k, const double2* __restrict__ tw, int twstep) {
    const double2 zk = a[k == m ? 0 : k], zc = a[k == 0 ? 0 : m - k];
    const double er = 0.5 * (zk.x + zc.x), ei = 0.5 * (zk.y - zc.y);
    const double orr = 0.5 * (zk.y + zc.y), oi = -0.5 * (zk.x - zc.x);
    const double2 w = k == m ? make_double2(-1.0, 0.0) : tw[k * twstep];
    return make_double2(er + w.x * orr - w.y * oi, ei + w.x * oi + w.y * orr);
}

// Frames of intervals shorter than the 0.1 s window (transform lengths below 1024): the one-wave kernel below lists them,
// a fixed grid of workgroups walks the list.
__global__ __launch_bounds__(256) void cepstrum_kernel(const Seg* __restrict__ segs, int max_seg, const int* __restrict__ hdr,
                                                       const double* __restrict__ res, int cap_res, int cap_frames,
                                                       const double* __restrict__ win1000, const double2* __restrict__ tw,
                                                       double preemph, double* __restrict__ ceps, const int* __restrict__ list,
                                                       const int* __restrict__ list_count, int list_cap) {
    __shared__ double2 a[NFFT_MAX / 2];
    __shared__ double xs[NFFT_MAX + 1];                 // windowed frame, then the ln-power half spectrum
    __shared__ double s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int count = min(*list_count, list_cap);       // (the producers drop what does not fit; the host sizes the list so that nothing is dropped)
    for (int item = blockIdx.x; item < count; item += gridDim.x) {
    __syncthreads();                                    // the previous frame's readers of a / xs / s_red are done
    const int clip = list[2 * item], f = list[2 * item + 1];
    const int nseg = hdr[4 * clip];
    const Seg* S = segs + (int64_t)clip * max_seg;
    const Seg s = S[find_seg<4>(S, nseg, f)];
    const int fl = f - (int)s.frame_off;
    const int nx = (int)s.nx, nfft = (int)s.nfft, m_out = (int)s.m_out;
    int log2n = 0;
    while ((1 << log2n) < nfft) ++log2n;
    const int m = nfft >> 1, log2m = log2n - 1, twstep = NFFT_MAX / nfft;
    const double* y = res + (int64_t)clip * cap_res + (int64_t)s.res_off;
    const double t = s.t1 + (double)fl * DT;
    const int64_t idx0 = (int64_t)floor((t - 0.5 * s.window - s.x1o) / DXO + 0.5);   // Sampled_xToNearestIndex, 0-based
    // gather with the pre-emphasis y[j] - a y[j-1] (the first sample of the sound is kept), frame mean
    double loc[4];
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = tid + 256 * r;
        double v = 0.0;
        if (i < nx) {
            const int64_t j = idx0 + i;
            if (j >= 0 && j < m_out) v = j >= 1 ? y[j] - preemph * y[j - 1] : y[j];
        }
        loc[r] = v;
        sum += v;
    }
    sum = wave_sum_f64(sum);
    if (lane == 0) s_red[wv] = sum;
    __syncthreads();
    const double mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (double)nx;
    const double imid = 0.5 * (nx + 1), edge = exp(-12.0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = tid + 256 * r;
        if (i < nfft) {
            double v = 0.0;
            if (i < nx) {
                double w;
                if (nx == 1000) w = win1000[i];
                else {
                    const double d = (double)(i + 1) - imid;
                    w = (exp(-48.0 * (d * d) / (double)((nx + 1) * (nx + 1))) - edge) / (1.0 - edge);
                }
                v = (loc[r] - mean) * w;
            }
            xs[i] = v;
        }
    }
    __syncthreads();
    // both transforms have real input (the second one even as well): one complex FFT of half the length each
    for (int j = tid; j < m; j += 256) a[bitrev(j, log2m)] = make_double2(xs[2 * j], xs[2 * j + 1]);
    __syncthreads();
    fft_inplace(a, m, log2m, tw, tid);
    // ln power of the bins 0..nfft/2 (spectrum scaled by the sample period)
    for (int k = tid; k <= m; k += 256) {
        const double2 X = real_fft_bin(a, m, k, tw, twstep);
        const double re = X.x * DXO, im = X.y * DXO;
        xs[k] = log(re * re + im * im + 1e-300);
    }
    __syncthreads();
    // real even sequence e[n] = lp[min(n, nfft - n)]
    for (int j = tid; j < m; j += 256) {
        const int n0 = 2 * j, n1 = 2 * j + 1;
        a[bitrev(j, log2m)] = make_double2(xs[n0 <= m ? n0 : nfft - n0], xs[n1 <= m ? n1 : nfft - n1]);
    }
    __syncthreads();
    fft_inplace(a, m, log2m, tw, tid);
    const double sdx = 1.0 / (DXO * (double)nfft);
    double* o = ceps + ((int64_t)clip * cap_frames + f) * NQ_MAX;
    for (int k = tid; k <= m; k += 256) {
        const double cv = real_fft_bin(a, m, k, tw, twstep).x * sdx;
        o[k] = cv * cv;
    }
    }
}

// One wavefront per frame for the usual case (0.1 s window = 1000 samples at 10 kHz, transform length 1024): both real
// transforms are 512-point complex transforms in registers (wave_fft.h, 8 points per lane), the conjugate pairs (k, 512 - k)
// of the real-input split are evaluated once each by the lane that holds k < 256, the ln-power half spectrum goes through
// LDS once to become the even sequence of the second transform.  No workgroup barrier; CEP_FRAMES frames per wave.
constexpr int CEP_FRAMES = 4;
constexpr int CEP_LDS_DOUBLES = wfft::Plan<8>::LDS_DOUBLES + 516;

__global__ __launch_bounds__(64, 4) void cepstrum_wave_kernel(const Seg* __restrict__ segs, int max_seg, const int* __restrict__ hdr,
                                                              const double* __restrict__ res, int cap_res, int cap_frames,
                                                              const double* __restrict__ win1000, const double2* __restrict__ tw,
                                                              double preemph, double* __restrict__ ceps, int* __restrict__ list,
                                                              int* __restrict__ list_count, int list_cap, int force_list) {
    using namespace wfft;
    __shared__ double lds[CEP_LDS_DOUBLES];
    double* lp = lds + Plan<8>::LDS_DOUBLES;            // ln-power half spectrum, bins 0 .. 512
    constexpr int R = 8, S = 512, H = 4;
    const int clip = blockIdx.y;
    const int nseg = hdr[4 * clip];
    if (nseg <= 0) return;
    const int nframes = hdr[4 * clip + 2];
    const int f0 = blockIdx.x * CEP_FRAMES;
    if (f0 >= nframes) return;
    const Seg* SG = segs + (int64_t)clip * max_seg;
    const int lane_ = threadIdx.x;
    LdsMem mem{lds};
    const int f1 = f0 + CEP_FRAMES < nframes ? f0 + CEP_FRAMES : nframes;
#pragma unroll 1
    for (int f = f0; f < f1; ++f) {
        int lane = lane_;                               // redefined per frame (keeps lane-only expressions out of the loop preheader)
        asm volatile("" : "+v"(lane));
        const Seg s = SG[find_seg<4>(SG, nseg, f)];
        if ((int)s.nfft != 1024 || force_list) {                      // a short interval: the workgroup kernel takes the frame
            if (lane == 0) {
                const int at = atomicAdd(list_count, 1);
                if (at < list_cap) { list[2 * at] = clip; list[2 * at + 1] = f; }
            }
            continue;
        }
        const int fl = f - (int)s.frame_off;
        const int nx = (int)s.nx, m_out = (int)s.m_out;
        const double* y = res + (int64_t)clip * cap_res + (int64_t)s.res_off;
        const double t = s.t1 + (double)fl * DT;
        const int idx0 = (int)floor((t - 0.5 * s.window - s.x1o) / DXO + 0.5);   // Sampled_xToNearestIndex, 0-based
        // gather with the pre-emphasis y[j] - a y[j-1] (the first sample of the sound is kept): element k = lane + 64 m of
        // the packed frame holds samples 2 k and 2 k + 1.  Loads are unconditional on clamped indices.
        cplx v[R];
        double sum = 0.0;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int i0 = 2 * (lane + 64 * m), j0 = idx0 + i0;
            const int ja = j0 - 1 < 0 ? 0 : (j0 - 1 > m_out - 1 ? m_out - 1 : j0 - 1);
            const int jb = j0 < 0 ? 0 : (j0 > m_out - 1 ? m_out - 1 : j0);
            const int jc = j0 + 1 < 0 ? 0 : (j0 + 1 > m_out - 1 ? m_out - 1 : j0 + 1);
            const double ya = y[ja], yb = y[jb], yc = y[jc];
            double e0 = 0.0, e1 = 0.0;
            if (i0 < nx && j0 >= 0 && j0 < m_out) e0 = j0 >= 1 ? yb - preemph * ya : yb;
            if (i0 + 1 < nx && j0 + 1 >= 0 && j0 + 1 < m_out) e1 = j0 + 1 >= 1 ? yc - preemph * yb : yc;
            v[m] = cplx{e0, e1};
            sum += e0 + e1;
        }
        const double mean = wave_sum_f64(sum) / (double)nx;
        const double imid = 0.5 * (nx + 1), edge = exp(-12.0);
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int i0 = 2 * (lane + 64 * m);
            double w0, w1;
            if (nx == 1000) {                           // uniform
                w0 = win1000[i0 < 999 ? i0 : 999];
                w1 = win1000[i0 + 1 < 999 ? i0 + 1 : 999];
            } else {
                const double d0 = (double)(i0 + 1) - imid, d1 = (double)(i0 + 2) - imid, q = (double)((nx + 1) * (nx + 1));
                w0 = (exp(-48.0 * (d0 * d0) / q) - edge) / (1.0 - edge);
                w1 = (exp(-48.0 * (d1 * d1) / q) - edge) / (1.0 - edge);
            }
            v[m] = cplx{i0 < nx ? (v[m].x - mean) * w0 : 0.0, i0 + 1 < nx ? (v[m].y - mean) * w1 : 0.0};
        }
        // first transform; X[k] = E + W^k O, X[512 - k] = conj(E - W^k O) from the pair (Z[k], Z[512 - k]); ln power
        wave_fft<R>(v, lds, lane, ld_c(tw, 2 * lane), ld_c(tw, (lane % 8) * 16));
        ac_spec_store<R>(v, mem, lane);
        wave_sync();
        const cplx wl = ld_c(tw, lane);
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const int k = lane + 64 * m;
            const int pp = k ? S / 2 - k : 0;
            cplx zc{mem.ld(pp), mem.ld(S / 2 + pp)};
            if (m == 0) zc = cplx{lane == 0 ? v[0].x : zc.x, lane == 0 ? v[0].y : zc.y};
            const cplx zk = v[m], w = mul_w64(wl, m * 4);                       // W_1024^(64 m) = W_64^(4 m)
            const double er = 0.5 * (zk.x + zc.x), ei = 0.5 * (zk.y - zc.y), orr = 0.5 * (zk.y + zc.y), oi = -0.5 * (zk.x - zc.x);
            const double tr = w.x * orr - w.y * oi, ti = w.x * oi + w.y * orr;
            const double ar = (er + tr) * DXO, ai = (ei + ti) * DXO, br = (er - tr) * DXO, bi = (ti - ei) * DXO;
            lp[k] = log(ar * ar + ai * ai + 1e-300);
            lp[S - k] = log(br * br + bi * bi + 1e-300);                        // k = 0: bin 512 = Re Z[0] - Im Z[0]
        }
        if (lane == 0) {                                                       // bin 256 pairs with itself: X = conj(Z[256])
            const double ar = v[H].x * DXO, ai = v[H].y * DXO;
            lp[S / 2] = log(ar * ar + ai * ai + 1e-300);
        }
        wave_sync();
        // the real even sequence e[n] = lp[min(n, 1024 - n)], packed as e[2 j] + i e[2 j + 1]
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int j = lane + 64 * m;
            v[m] = m < H ? cplx{lp[2 * j], lp[2 * j + 1]} : cplx{lp[1024 - 2 * j], lp[1023 - 2 * j]};
        }
        wave_sync();
        wave_fft<R>(v, lds, lane, ld_c(tw, 2 * lane), ld_c(tw, (lane % 8) * 16));
        ac_spec_store<R>(v, mem, lane);
        wave_sync();
        const double sdx = 1.0 / (DXO * 1024.0);
        double* o = ceps + ((int64_t)clip * cap_frames + f) * NQ_MAX;
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const int k = lane + 64 * m;
            const int pp = k ? S / 2 - k : 0;
            cplx zc{mem.ld(pp), mem.ld(S / 2 + pp)};
            if (m == 0) zc = cplx{lane == 0 ? v[0].x : zc.x, lane == 0 ? v[0].y : zc.y};
            const cplx zk = v[m], w = mul_w64(wl, m * 4);
            const double er = 0.5 * (zk.x + zc.x), orr = 0.5 * (zk.y + zc.y), oi = -0.5 * (zk.x - zc.x);
            const double tr = w.x * orr - w.y * oi;
            const double ca = (er + tr) * sdx, cb = (er - tr) * sdx;
            o[k] = ca * ca;
            o[S - k] = cb * cb;
        }
        if (lane == 0) {
            const double cm = v[H].x * sdx;
            o[S / 2] = cm * cm;
        }
        wave_sync();                                    // the next frame rewrites the exchange buffer
    }
}

// ---- 4. smoothed cepstral peak prominence of every frame --------------------------------------------------
// ascending bitonic sort of n (power of two) doubles in LDS by 256 threads
__device__ void bitonic_sort(double* v, int n, int tid) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const double x = v[i], y = v[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { v[i] = y; v[l] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// The same sort for n = 256 or 512 with the elements in registers (thread t owns v[t] and, for 512, v[t + 256]): the
// compare-exchange distances below 64 are wave shuffles, 256 is thread-local, only 64 and 128 go through LDS.
template <int E>
__device__ void bitonic_sort_reg(double* v, int tid) {
    constexpr int N = 256 * E;
    double x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = v[tid + 256 * e];
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j == 256) {                                   // E == 2, k == 512: partner in the same thread, ascending
                if (E == 2) {
                    const double lo = x[0] < x[E - 1] ? x[0] : x[E - 1], hi = x[0] < x[E - 1] ? x[E - 1] : x[0];
                    x[0] = lo; x[E - 1] = hi;
                }
                continue;
            }
            double y[E];
            if (j >= 64) {
                __syncthreads();                               // earlier readers of v are done
#pragma unroll
                for (int e = 0; e < E; ++e) v[tid + 256 * e] = x[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) y[e] = v[(tid ^ j) + 256 * e];
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) y[e] = __shfl_xor(x[e], j, 64);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = tid + 256 * e;
                const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
                const double mn = x[e] < y[e] ? x[e] : y[e], mx = x[e] < y[e] ? y[e] : x[e];
                x[e] = keep_min ? mn : mx;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) v[tid + 256 * e] = x[e];
    __syncthreads();
}

__device__ __forceinline__ void sort_lds(double* v, int n, int tid) {
    if (n == 512) bitonic_sort_reg<2>(v, tid);
    else if (n == 256) bitonic_sort_reg<1>(v, tid);
    else bitonic_sort(v, n, tid);
}

__device__ double quantile_half(const double* a, int n) {     // Praat NUMquantile(sorted, 0.5)
    if (n < 1) return 0.0;
    if (n == 1) return a[0];
    const double place = 0.5 * n + 0.5;
    int left = (int)floor(place);
    left = left < 1 ? 1 : (left > n - 1 ? n - 1 : left);
    if (a[left] == a[left - 1]) return a[left - 1];
    return a[left - 1] + (place - left) * (a[left] - a[left - 1]);
}

__global__ __launch_bounds__(256) void cpp_frame_kernel(const Seg* __restrict__ segs, int max_seg, const int* __restrict__ hdr,
                                                        const double* __restrict__ ceps, int cap_frames, int n_time,
                                                        int n_quef, double pitch_floor, double pitch_ceiling,
                                                        double* __restrict__ cpp_out, const int* __restrict__ list,
                                                        const int* __restrict__ list_count, int list_cap) {
    __shared__ double zt[NQ_MAX + 3], db[NQ_MAX + 3], srt[NFFT_MAX];
    __shared__ double s_val[4];
    __shared__ int s_ord[4];
    __shared__ double s_xq[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int count = min(*list_count, list_cap);       // (the producers drop what does not fit; the host sizes the list so that nothing is dropped)
    for (int item = blockIdx.x; item < count; item += gridDim.x) {
    __syncthreads();                                    // the previous frame's readers of the shared arrays are done
    const int clip = list[2 * item], f = list[2 * item + 1];
    const int nseg = hdr[4 * clip];
    const Seg* S = segs + (int64_t)clip * max_seg;
    const Seg s = S[find_seg<4>(S, nseg, f)];
    const int fl = f - (int)s.frame_off, nf = (int)s.nf, nfft = (int)s.nfft, nq = nfft / 2 + 1;
    const double* Z = ceps + ((int64_t)clip * cap_frames + (int64_t)s.frame_off) * NQ_MAX;   // frames of this interval
    // moving average over time (VECsmoothByMovingAverage: [i - w/2, i + w/2], one less on the right for even w)
    int lo = fl, hi = fl;
    if (n_time > 1) {
        lo = fl - n_time / 2;
        hi = fl + n_time / 2 - ((n_time & 1) == 0 ? 1 : 0);
        lo = lo < 0 ? 0 : lo;
        hi = hi > nf - 1 ? nf - 1 : hi;
    }
    for (int q = tid; q < nq; q += 256) {
        double v = 0.0;
        for (int j = lo; j <= hi; ++j) v += Z[(int64_t)j * NQ_MAX + q];
        zt[q] = n_time > 1 ? v / (double)(hi - lo + 1) : v;
    }
    __syncthreads();
    // moving average over quefrency, then dB
    for (int q = tid; q < nq; q += 256) {
        double v = zt[q];
        if (n_quef > 1) {
            int a = q - n_quef / 2, b = q + n_quef / 2 - ((n_quef & 1) == 0 ? 1 : 0);
            a = a < 0 ? 0 : a;
            b = b > nq - 1 ? nq - 1 : b;
            v = 0.0;
            for (int j = a; j <= b; ++j) v += zt[j];
            v /= (double)(b - a + 1);
        }
        db[q] = 10.0 * log10(v + 1e-30);
    }
    __syncthreads();
    // Theil's incomplete method over all nq points: slope = median of the nc half-distance slopes ...
    const int nc = nq / 2, n2 = (nq & 1) ? nc + 1 : nc;
    int p2 = 1;
    while (p2 < nc) p2 <<= 1;
    for (int i = tid; i < p2; i += 256)
        srt[i] = i < nc ? (db[n2 + i] - db[i]) / ((double)(n2 + i) * DQ - (double)i * DQ) : INFINITY;
    __syncthreads();
    sort_lds(srt, p2, tid);
    const double slope = quantile_half(srt, nc);
    __syncthreads();
    // ... intercept = median of the residual offsets.  nq = 2^k + 1 (the usual 513): sort the first 2^k values
    // and place the last one by comparison, which halves the network; NUMquantile(0.5) of an odd count is the
    // middle order statistic.
    double icpt;
    p2 = 1;
    while (p2 < nq) p2 <<= 1;
    if (nq == (p2 >> 1) + 1 && nq >= 3) {
        const int hn = p2 >> 1;                                  // nq - 1
        for (int i = tid; i < hn; i += 256) srt[i] = db[i] - slope * ((double)i * DQ);
        __syncthreads();
        sort_lds(srt, hn, tid);
        const double e = db[nq - 1] - slope * ((double)(nq - 1) * DQ);
        const double lo_v = srt[hn / 2 - 1], hi_v = srt[hn / 2];  // the middle of the union is the median of (lo, e, hi)
        icpt = e <= lo_v ? lo_v : (e >= hi_v ? hi_v : e);
    } else {
        for (int i = tid; i < p2; i += 256) srt[i] = i < nq ? db[i] - slope * ((double)i * DQ) : INFINITY;
        __syncthreads();
        sort_lds(srt, p2, tid);
        icpt = quantile_half(srt, nq);
    }
    // Vector_getMaximumAndX (parabolic) over [1/ceiling, 1/floor]: end points first, then the local maxima in
    // ascending order, a later candidate wins only if strictly greater -> (value, order) reduction
    const double qlo = 1.0 / pitch_ceiling, qhi = 1.0 / pitch_floor;
    int imin = (int)ceil(qlo / DQ), imax = (int)floor(qhi / DQ);
    imax = imax > nq - 1 ? nq - 1 : imax;
    double best = -INFINITY, bx = 0.0;
    int bord = 0x7fffffff;
    if (imax >= imin) {
        if (tid == 0) { best = db[imin]; bx = (double)imin; bord = 0; }
        if (tid == 1 && db[imax] > db[imin]) { best = db[imax]; bx = (double)imax; bord = 1; }
        const int a = imin < 1 ? 1 : imin, b = imax > nq - 2 ? nq - 2 : imax;
        for (int i = a + tid; i <= b; i += 256) {
            if (db[i] > db[i - 1] && db[i] >= db[i + 1]) {
                const double dy = 0.5 * (db[i + 1] - db[i - 1]), d2y = 2.0 * db[i] - db[i - 1] - db[i + 1];
                const double v = db[i] + 0.5 * dy * dy / d2y;
                if (v > best) { best = v; bx = (double)i + dy / d2y; bord = 2 + i; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(best, o, 64), ox = __shfl_xor(bx, o, 64);
        const int oo = __shfl_xor(bord, o, 64);
        if (ov > best || (ov == best && oo < bord)) { best = ov; bx = ox; bord = oo; }
    }
    if (lane == 0) { s_val[wv] = best; s_xq[wv] = bx; s_ord[wv] = bord; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_val[w] > best || (s_val[w] == best && s_ord[w] < bord)) { best = s_val[w]; bx = s_xq[w]; bord = s_ord[w]; }
        double out = __longlong_as_double(0x7ff8000000000000LL);
        if (imax >= imin) {
            double qpeak = bx * DQ;
            qpeak = qpeak < qlo ? qlo : (qpeak > qhi ? qhi : qpeak);
            out = best - (slope * qpeak + icpt);
        }
        cpp_out[(int64_t)clip * cap_frames + f] = out;
    }
    }
}

// One wavefront per frame for the frames of full-window intervals (513 quefrency bins: lane l holds bins 8 l .. 8 l + 7,
// lane 63 also bin 512).  The two medians of Theil's line are order statistics, not sorts: wave_select_pair finds the
// elements of rank t and t + 1 of the 64 E values the lanes hold by partitioning around pivots taken from the data
// (compare + ballot + scalar popcount per element and round, about a dozen rounds), with no exchange of data between lanes
// and no arithmetic on the values, so the medians are the ones a full sort gives.  Same operation order as the
// workgroup kernel everywhere else (fp contraction is off in this file): the two kernels return identical bits.
template <int E>
__device__ __forceinline__ void wave_select_pair(const double (&x)[E], int t, double& a, double& b) {
    double lo = -INFINITY, hi = INFINITY;               // the element of rank t lies in the open interval (lo, hi)
    int count_le = 0;
    a = 0.0;
    for (int round = 0; round <= 64 * E; ++round) {
        double p = 0.0;
        bool found = false;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (!found) {
                const unsigned long long mk = __ballot(x[e] > lo && x[e] < hi);
                if (mk) {
                    const int src = __ffsll((long long)mk) - 1;
                    p = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x[e]), src),
                                         __builtin_amdgcn_readlane(__double2loint(x[e]), src));
                    found = true;
                }
            }
        }
        if (!found) break;                              // cannot happen: rank t always has an element in the interval
        int lt = 0, le = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            lt += __popcll(__ballot(x[e] < p));
            le += __popcll(__ballot(x[e] <= p));
        }
        if (lt <= t && t < le) { a = p; count_le = le; break; }
        if (t < lt) hi = p; else lo = p;
    }
    if (count_le > t + 1) { b = a; return; }
    double mn = INFINITY;
#pragma unroll
    for (int e = 0; e < E; ++e) mn = (x[e] > a && x[e] < mn) ? x[e] : mn;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(mn, o, 64);
        mn = ov < mn ? ov : mn;
    }
    b = mn;
}

constexpr int CPPF_FRAMES = 4;

__global__ __launch_bounds__(64, 4) void cpp_frame_wave_kernel(const Seg* __restrict__ segs, int max_seg, const int* __restrict__ hdr,
                                                               const double* __restrict__ ceps, int cap_frames, int n_time,
                                                               int n_quef, double pitch_floor, double pitch_ceiling,
                                                               double* __restrict__ cpp_out, int* __restrict__ list,
                                                               int* __restrict__ list_count, int list_cap, int force_list) {
    __shared__ double sq[NQ_MAX + 7];                    // smoothed power per bin, then dB per bin
    constexpr int NQ = NQ_MAX;                           // 513
    const int clip = blockIdx.y;
    const int nseg = hdr[4 * clip];
    if (nseg <= 0) return;
    const int nframes = hdr[4 * clip + 2];
    const int f0 = blockIdx.x * CPPF_FRAMES;
    if (f0 >= nframes) return;
    const Seg* SG = segs + (int64_t)clip * max_seg;
    const int lane_ = threadIdx.x;
    const int f1 = f0 + CPPF_FRAMES < nframes ? f0 + CPPF_FRAMES : nframes;
    const double qlo = 1.0 / pitch_ceiling, qhi = 1.0 / pitch_floor;
#pragma unroll 1
    for (int f = f0; f < f1; ++f) {
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        const Seg s = SG[find_seg<4>(SG, nseg, f)];
        if ((int)s.nfft != 1024 || force_list) {                      // a short interval: the workgroup kernel takes the frame
            if (lane == 0) {
                const int at = atomicAdd(list_count, 1);
                if (at < list_cap) { list[2 * at] = clip; list[2 * at + 1] = f; }
            }
            continue;
        }
        const int fl = f - (int)s.frame_off, nf = (int)s.nf;
        const double* Z = ceps + ((int64_t)clip * cap_frames + (int64_t)s.frame_off) * NQ_MAX;   // frames of this interval
        // moving average over time (VECsmoothByMovingAverage: [i - w/2, i + w/2], one less on the right for even w)
        int lo = fl, hi = fl;
        if (n_time > 1) {
            lo = fl - n_time / 2;
            hi = fl + n_time / 2 - ((n_time & 1) == 0 ? 1 : 0);
            lo = lo < 0 ? 0 : lo;
            hi = hi > nf - 1 ? nf - 1 : hi;
        }
        const int q0 = 8 * lane;
        double zt[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) zt[e] = 0.0;
        for (int j = lo; j <= hi; ++j) {
            const double* row = Z + (int64_t)j * NQ_MAX;
            double r[9];
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = row[q0 + e];
            r[8] = row[NQ - 1];                          // bin 512 (lane 63's ninth)
#pragma unroll
            for (int e = 0; e < 9; ++e) zt[e] += r[e];
        }
        if (n_time > 1) {
            const double cnt = (double)(hi - lo + 1);
#pragma unroll
            for (int e = 0; e < 9; ++e) zt[e] = zt[e] / cnt;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sq[q0 + e] = zt[e];
        if (lane == 63) sq[NQ - 1] = zt[8];
        wfft::wave_sync();
        // moving average over quefrency, then dB
        double db[9];
        if (n_quef == 10) {                              // [q - 5, q + 4]: 18 consecutive values cover the lane's nine bins
            double w[18];
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const int jj = q0 - 5 + i;
                const double v = sq[jj < 0 ? 0 : (jj > NQ - 1 ? NQ - 1 : jj)];
                w[i] = (jj >= 0 && jj <= NQ - 1) ? v : 0.0;
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const int q = q0 + e;
                int a = q - 5, b = q + 4;
                a = a < 0 ? 0 : a;
                b = b > NQ - 1 ? NQ - 1 : b;
                double v = 0.0;
#pragma unroll
                for (int i = 0; i < 10; ++i) v += w[e + i];   // entries outside [0, 512] are 0.0: the sum is the one over [a, b]
                v /= (double)(b - a + 1);
                db[e] = 10.0 * log10(v + 1e-30);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const int q = q0 + e < NQ ? q0 + e : NQ - 1;
                double v = sq[q];
                if (n_quef > 1) {
                    int a = q - n_quef / 2, b = q + n_quef / 2 - ((n_quef & 1) == 0 ? 1 : 0);
                    a = a < 0 ? 0 : a;
                    b = b > NQ - 1 ? NQ - 1 : b;
                    v = 0.0;
                    for (int j = a; j <= b; ++j) v += sq[j];
                    v /= (double)(b - a + 1);
                }
                db[e] = 10.0 * log10(v + 1e-30);
            }
        }
        wfft::wave_sync();
#pragma unroll
        for (int e = 0; e < 8; ++e) sq[q0 + e] = db[e];
        if (lane == 63) sq[NQ - 1] = db[8];
        wfft::wave_sync();
        // Theil's incomplete method over all 513 points: slope = median of the 256 half-distance slopes ...
        double sl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * lane + e;
            sl[e] = (sq[257 + i] - sq[i]) / ((double)(257 + i) * DQ - (double)i * DQ);
        }
        double m0, m1;
        wave_select_pair<4>(sl, 127, m0, m1);                                   // NUMquantile(0.5) of 256: place 128.5
        const double slope = m1 == m0 ? m0 : m0 + 0.5 * (m1 - m0);
        // ... intercept = median of the 513 residual offsets: the ranks 255 and 256 of the first 512 bracket it
        double rs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) rs[e] = db[e] - slope * ((double)(q0 + e) * DQ);
        double lo_v, hi_v;
        wave_select_pair<8>(rs, 255, lo_v, hi_v);
        const double e512 = sq[NQ - 1] - slope * ((double)(NQ - 1) * DQ);
        const double icpt = e512 <= lo_v ? lo_v : (e512 >= hi_v ? hi_v : e512);
        // Vector_getMaximumAndX (parabolic) over [1/ceiling, 1/floor]: end points first, then the local maxima in
        // ascending order, a later candidate wins only if strictly greater -> (value, order) reduction
        int imin = (int)ceil(qlo / DQ), imax = (int)floor(qhi / DQ);
        imax = imax > NQ - 1 ? NQ - 1 : imax;
        double best = -INFINITY, bx = 0.0;
        int bord = 0x7fffffff;
        if (imax >= imin) {
            if (lane == 0) { best = sq[imin]; bx = (double)imin; bord = 0; }
            if (lane == 1 && sq[imax] > sq[imin]) { best = sq[imax]; bx = (double)imax; bord = 1; }
            const int a = imin < 1 ? 1 : imin, b = imax > NQ - 2 ? NQ - 2 : imax;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = q0 + e;
                const int il = i >= 1 ? i - 1 : 0, ir = i + 1 <= NQ - 1 ? i + 1 : NQ - 1;
                const double dl = sq[il], dc = db[e], dr = sq[ir];
                if (i >= a && i <= b && dc > dl && dc >= dr) {
                    const double dy = 0.5 * (dr - dl), d2y = 2.0 * dc - dl - dr;
                    const double v = dc + 0.5 * dy * dy / d2y;
                    if (v > best) { best = v; bx = (double)i + dy / d2y; bord = 2 + i; }
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double ov = __shfl_xor(best, o, 64), ox = __shfl_xor(bx, o, 64);
            const int oo = __shfl_xor(bord, o, 64);
            if (ov > best || (ov == best && oo < bord)) { best = ov; bx = ox; bord = oo; }
        }
        if (lane == 0) {
            double out = __longlong_as_double(0x7ff8000000000000LL);
            if (imax >= imin) {
                double qpeak = bx * DQ;
                qpeak = qpeak < qlo ? qlo : (qpeak > qhi ? qhi : qpeak);
                out = best - (slope * qpeak + icpt);
            }
            cpp_out[(int64_t)clip * cap_frames + f] = out;
        }
        wfft::wave_sync();                              // the next frame rewrites sq
    }
}

// ---- 5. CPPS per interval, mean of the values above 4 dB per clip -----------------------------------------
__global__ __launch_bounds__(256) void reduce_kernel(const Seg* __restrict__ segs, int max_seg, const int* __restrict__ hdr,
                                                     const double* __restrict__ cpp_frames, int cap_frames, double threshold,
                                                     double* __restrict__ out) {
    __shared__ double s_red[4];
    const int clip = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nseg = hdr[4 * clip], fail = hdr[4 * clip + 1];
    const Seg* S = segs + (int64_t)clip * max_seg;
    const double* F = cpp_frames + (int64_t)clip * cap_frames;
    double total = 0.0;
    int kept = 0;
    for (int k = 0; k < nseg; ++k) {
        const int off = (int)S[k].frame_off, nf = (int)S[k].nf;
        double v = 0.0;
        for (int i = tid; i < nf; i += 256) v += F[off + i];
        v = wave_sum_f64(v);
        __syncthreads();
        if (lane == 0) s_red[wv] = v;
        __syncthreads();
        const double cpps = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (double)nf;
        if (cpps == cpps && cpps > threshold) { total += cpps; ++kept; }        // :293
    }
    if (tid == 0) out[clip] = (kept > 0 && !fail) ? total / (double)kept : __longlong_as_double(0x7ff8000000000000LL);
}

}  // namespace mshds_cpp
}  // namespace rsaf

using namespace rsaf;
using namespace rsaf::mshds_cpp;

extern "C" {

int rsaf_mshds_cpp_seg_doubles(void) { return SEG_DOUBLES; }

int rsaf_mshds_cpp(const float* wav, const void* clip_info, int n_clips, const double* pulses, int max_pulses,
                   const int* n_pulses, const double* window1000, const double* twiddle1024, int max_seg, int cap_res,
                   int cap_frames, void* seg_table, int* hdr, double* resampled, double* cepstrogram, double* cpp_frames,
                   double* lowpassed, int64_t lp_origin, void* lp_work, int64_t cap_work, int lg_max, double* out,
                   rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && n_clips <= 65535 && max_pulses >= 0, "bad clip/pulse count");
    if (n_clips == 0) return RSAF_OK;
    RSAF_CHECK_ARG(wav && clip_info && pulses && n_pulses && window1000 && twiddle1024 && seg_table && hdr && resampled &&
                   cepstrogram && cpp_frames && lowpassed && lp_work && out, "NULL pointer");
    RSAF_CHECK_ARG(max_seg >= 1 && cap_res >= 1 && cap_frames >= 1 && cap_frames <= 65535 * 32 && cap_work >= 1024 &&
                   lg_max >= 11 && lg_max <= resample::LP_LG_MAX, "bad capacities");
    hipStream_t s = (hipStream_t)stream;
    const ClipInfo* ci = (const ClipInfo*)clip_info;
    Seg* segs = (Seg*)seg_table;
    {
        ProfScope prof("mshds_cpp_segments", s, 0.0, 0.0);
        hipLaunchKernelGGL(segments_kernel, dim3(n_clips), dim3(64), 0, s, ci, pulses, max_pulses, n_pulses, 0.02, 0.1, 60.0,
                           segs, max_seg, cap_res, cap_frames, cap_work, lg_max, hdr);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    {
        ProfScope prof("mshds_cpp_resample", s, 0.0, 0.0);
        resample::LpTables T;
        const int rc = resample::lp_tables((int64_t)1 << lg_max, &T);
        if (rc != RSAF_OK) return rc;
        T.lg_max = lg_max;
        auto lds_for = [](int lg_a, int lg_b) {
            size_t lds = 0;
            for (int lg = lg_a; lg <= lg_b; ++lg) {
                const resample::LpGeom g = resample::lp_geom(lg);
                lds = std::max(lds, std::max(((size_t)g.C << g.log1), ((size_t)2 << g.log2)) * sizeof(resample::c64));
            }
            return lds;
        };
        constexpr int LG_LDS = 14;                                           // 2^13 complex numbers = 128 KB of LDS
        resample::c64* wk = (resample::c64*)lp_work;
        {
            const int lg_hi = std::min(LG_LDS, lg_max);
            const size_t lds = ((size_t)1 << (lg_hi - 1)) * sizeof(resample::c64);
            if (lds > 48 * 1024)
                RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)seg_lowpass_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(seg_lowpass_lds_kernel, dim3(32, n_clips), dim3(1024), lds, s, wav, ci, segs, max_seg, hdr, lowpassed,
                               lp_origin, lg_hi, T);
        }
        if (lg_max > LG_LDS) {
            const int lg_lo = LG_LDS + 1, lg_hi = lg_max;
            const size_t lds = lds_for(lg_lo, lg_hi);
            if (lds > 48 * 1024) {
                RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)seg_lowpass_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)seg_lowpass_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)seg_lowpass_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            const dim3 grid(64, n_clips);                                    // workgroups that walk the items of a clip
            hipLaunchKernelGGL(seg_lowpass_kernel<0>, grid, dim3(256), lds, s, wav, ci, segs, max_seg, hdr, wk, cap_work, lowpassed, lp_origin, lg_lo, lg_hi, T);
            hipLaunchKernelGGL(seg_lowpass_kernel<1>, grid, dim3(256), lds, s, wav, ci, segs, max_seg, hdr, wk, cap_work, lowpassed, lp_origin, lg_lo, lg_hi, T);
            hipLaunchKernelGGL(seg_lowpass_kernel<2>, grid, dim3(256), lds, s, wav, ci, segs, max_seg, hdr, wk, cap_work, lowpassed, lp_origin, lg_lo, lg_hi, T);
        }
        const char* pe = getenv("RSAF_CPP_POLYPHASE");
        hipLaunchKernelGGL(resample_kernel, dim3((cap_res + 255) / 256, n_clips), dim3(256), 0, s, (const double*)lowpassed, lp_origin,
                           ci, segs, max_seg, hdr, cap_res, resampled, (pe && pe[0] == '0') ? 0 : 1);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    int force_list = 0;
    // RSAF_CPP_WAVE: "0" = both per-frame kernels in their workgroup form, "c" = only the cepstrum by one wave per frame
    // (bit 0: cepstrum through the list, bit 1: smoothed-CPP frames through the list)
    { const char* e = getenv("RSAF_CPP_WAVE"); force_list = !e ? 0 : (e[0] == '0' ? 3 : (e[0] == 'c' ? 2 : 0)); }
    {
        ProfScope prof("mshds_cpp_cepstrum", s, 0.0, 0.0);
        // frames with the full 1024-point transform: one wave each; the others are listed in lp_work (free once the intervals
        // are resampled; the counter is its last slot) and taken by the workgroup kernel
        const int64_t lp_doubles = (int64_t)n_clips * cap_work * 2;
        int* list = reinterpret_cast<int*>(lp_work);
        int* list_count = list + 2 * (lp_doubles - 1);
        const int list_cap = (int)std::min<int64_t>(lp_doubles - 1, 0x7fffffff);
        // every frame of every clip may be listed at most once: the list holds them all, nothing can be dropped
        RSAF_CHECK_ARG((int64_t)n_clips * cap_frames <= list_cap, "frame list smaller than the frame count");
        RSAF_CHECK_HIP(hipMemsetAsync(list_count, 0, sizeof(int), s));
        const double pre = exp(-2.0 * PI * 50.0 * DXO);
        hipLaunchKernelGGL(cepstrum_wave_kernel, dim3((cap_frames + CEP_FRAMES - 1) / CEP_FRAMES, n_clips), dim3(64), 0, s, segs,
                           max_seg, hdr, resampled, cap_res, cap_frames, window1000, (const double2*)twiddle1024, pre, cepstrogram,
                           list, list_count, list_cap, force_list & 1);
        RSAF_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL(cepstrum_kernel, dim3(2048), dim3(256), 0, s, segs, max_seg, hdr, resampled, cap_res, cap_frames,
                           window1000, (const double2*)twiddle1024, pre, cepstrogram, list, list_count, list_cap);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    {
        ProfScope prof("mshds_cpp_frames", s, 0.0, 0.0);
        const int64_t lp_doubles = (int64_t)n_clips * cap_work * 2;
        int* list = reinterpret_cast<int*>(lp_work);
        int* list_count = list + 2 * (lp_doubles - 1);
        const int list_cap = (int)std::min<int64_t>(lp_doubles - 1, 0x7fffffff);
        RSAF_CHECK_HIP(hipMemsetAsync(list_count, 0, sizeof(int), s));
        hipLaunchKernelGGL(cpp_frame_wave_kernel, dim3((cap_frames + CPPF_FRAMES - 1) / CPPF_FRAMES, n_clips), dim3(64), 0, s, segs,
                           max_seg, hdr, cepstrogram, cap_frames, (int)floor(0.01 / DT), (int)floor(0.001 / DQ), 60.0, 330.0,
                           cpp_frames, list, list_count, list_cap, force_list & 2);
        RSAF_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL(cpp_frame_kernel, dim3(2048), dim3(256), 0, s, segs, max_seg, hdr, cepstrogram, cap_frames,
                           (int)floor(0.01 / DT), (int)floor(0.001 / DQ), 60.0, 330.0, cpp_frames, list, list_count, list_cap);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(reduce_kernel, dim3(n_clips), dim3(256), 0, s, segs, max_seg, hdr, cpp_frames, cap_frames, 4.0, out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

}  // extern "C"
