// Sample-rate conversion to 16 kHz in front of the extractors for gfx950.
//   rsaf_resample_sinc_hann : torchaudio.transforms.Resample defaults (sinc_interp_hann, width 6, rolloff 0.99),
//                             the resampler of src/foundation_model_extractor.py:93-94.  Polyphase FIR in fp32,
//                             one thread per output sample; the host passes each phase's non-zero taps.
//   rsaf_resample_praat     : Sound.resample(16000, 50) of src/mshds_extractor.py:419 as Praat does it: when the rate
//                             goes down, a brick-wall low-pass of the whole sound by a real FFT of the first power of
//                             two >= n + 2000 samples (fp64, four-step transform through LDS), then NUM_interpolate_sinc
//                             of the given depth on the re-centred sample grid, fp64 math.
// The FIR kernels are HBM-bound streaming kernels: 4 B read per input sample (taps and neighbours come from L1/L2), 4 B
// written.  The FFT low-pass moves 16 B per complex point and pass (3 passes forward + back over the work buffer).
#include <hip/hip_runtime.h>

#include <cmath>
#include <algorithm>
#include <cstdint>
#include <map>
#include <mutex>
#include <vector>

#include "praat_interp.h"
#include "rsaf_common.h"

namespace rsaf {
namespace resample {

__global__ __launch_bounds__(256) void sinc_hann_kernel(const float* __restrict__ x, int64_t n_in, const float* __restrict__ taps,
                                                        const int* __restrict__ tap_start, int n_phase, int orig,
                                                        int taps_per_phase, float* __restrict__ out, int64_t n_out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const int64_t i = o / n_phase;
    const int p = (int)(o - i * n_phase);
    const float* w = taps + (int64_t)p * taps_per_phase;
    const int64_t j0 = i * orig + tap_start[p];
    float acc = 0.0f;
    for (int k = 0; k < taps_per_phase; ++k) {
        const int64_t j = j0 + k;
        const float v = (j >= 0 && j < n_in) ? x[j] : 0.0f;
        acc = fmaf(w[k], v, acc);
    }
    out[o] = acc;
}

constexpr double PI = 3.14159265358979323846;

// ---- whole-sound FFT low-pass (Praat Sound_resample, anti-aliasing branch) -------------------------------------
// The nfft real samples (1 000 zeros, the sound, zeros) are the M = nfft / 2 complex numbers z[n] = (d[2n], d[2n+1]),
// n = n1 N2 + n2.  Forward: column transforms over n1 (decimation in frequency, in place in LDS, output row r holds
// k1 = bitrev(r)), times W_M^(n2 k1); then row transforms over n2 (same scheme, LDS position p holds k2 = bitrev(p)):
// Z[k1 + N1 k2].  The real-transform bins X[k], X[M - k] come from Z[k], Z[M - k], which live in rows k1 and N1 - k1: one
// workgroup owns both rows, clears what Praat clears, folds back to Z' and runs the inverse row transforms (decimation
// in time: bit-reversed in, natural out).  The inverse column pass undoes the first one.  No pass reorders memory.
typedef double2 c64;
constexpr int TW_LOG = 12, TW_N = 1 << TW_LOG;        // W_4096^j: butterflies of every LDS transform (length <= 4096)
constexpr int ANTI_TURN_AROUND = 1000;

__device__ __forceinline__ c64 cmul(c64 a, c64 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ c64 cmulc(c64 a, c64 b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a conj(b)
__device__ __forceinline__ c64 cadd(c64 a, c64 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c64 csub(c64 a, c64 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c64 cconj(c64 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ c64 mul_mi(c64 a) { return make_double2(a.y, -a.x); }    // a * (-i)
__device__ __forceinline__ c64 mul_pi(c64 a) { return make_double2(-a.y, a.x); }    // a * (+i)
__device__ __forceinline__ int bitrev(int v, int bits) { return bits ? (int)(__brev((unsigned)v) >> (32 - bits)) : 0; }

struct LpTables {
    const c64* tw;      // [4096]  e^(-2 pi i j / 4096)
    const c64* lo;      // [4096]  e^(-2 pi i j / nfft_max)
    const c64* hi;      // [nfft_max / 4096 or 1]  e^(-2 pi i 4096 j / nfft_max)
    int lg_max;         // the tables belong to nfft_max = 2^lg_max; a shorter transform strides through them
};

// one sound of a batch: n samples at in + in_off -> out + out_off, transform of 2^lg samples in work + work_off
struct LpSig {
    int64_t in_off, out_off, work_off;
    int n, lg;
};
static_assert(sizeof(LpSig) == 32, "LpSig layout");
struct LpBatch {
    const LpSig* sigs;  // device array indexed by blockIdx.y, or nullptr: `one`
    LpSig one;
    double upfactor;
};

struct LpGeom { int log1, log2, C; };
__host__ __device__ inline LpGeom lp_geom(int lg) {
    LpGeom g;
    const int logM = lg - 1;                            // lg >= 11
    g.log2 = logM - 1 < 11 ? logM - 1 : 11;             // rows of at most 2 048 points, at least 2 rows
    g.log1 = logM - g.log2;
    g.C = 8;
    while (g.C > 1 && ((int64_t)g.C << g.log1) > 4096) g.C >>= 1;
    return g;
}

// e^(-2 pi i p / 2^lg), 0 <= p < 2^lg
__device__ __forceinline__ c64 w_nfft(const LpTables& T, int64_t p, int lg) {
    p <<= T.lg_max - lg;
    return cmul(T.lo[p & (TW_N - 1)], T.hi[p >> TW_LOG]);
}

// nseq interleaved sequences of length 2^logL in LDS (element e of sequence q at buf[e * es + q * ss]); INV = false:
// decimation in frequency, forward twiddles, natural in / bit-reversed out; INV = true: decimation in time, conjugate
// twiddles, bit-reversed in / natural out, unnormalised.  Two butterfly layers per barrier.
template <bool INV>
__device__ void lds_fft(c64* buf, int logL, int nseq, int es, int ss, const c64* __restrict__ tw) {
    const int L = 1 << logL;
    if (INV && (logL & 1)) {
        for (int t = threadIdx.x; t < nseq * (L >> 1); t += blockDim.x) {
            const int q = t / (L >> 1), u = t - q * (L >> 1);
            c64* p = buf + q * ss + (2 * u) * es;
            const c64 a = p[0], b = p[es];
            p[0] = cadd(a, b);
            p[es] = csub(a, b);
        }
        __syncthreads();
    }
    const int first = INV ? (2 + (logL & 1)) : logL, last = INV ? logL : (2 + (logL & 1));
    for (int sl = first; INV ? sl <= last : sl >= last; sl += INV ? 2 : -2) {
        const int ql = sl - 2, qn = 1 << ql;
        for (int t = threadIdx.x; t < nseq * (L >> 2); t += blockDim.x) {
            const int q = t / (L >> 2), u = t - q * (L >> 2);
            const int blk = u >> ql, j = u & (qn - 1);
            c64* p = buf + q * ss + ((blk << sl) + j) * es;
            const int st = qn * es;
            const c64 w1 = tw[j << (TW_LOG - sl)], w2 = tw[(2 * j) << (TW_LOG - sl)];
            const c64 a0 = p[0], a1 = p[st], a2 = p[2 * st], a3 = p[3 * st];
            if (!INV) {
                const c64 b0 = cadd(a0, a2), b2 = cmul(csub(a0, a2), w1);
                const c64 b1 = cadd(a1, a3), b3 = cmul(mul_mi(csub(a1, a3)), w1);
                p[0] = cadd(b0, b1);
                p[st] = cmul(csub(b0, b1), w2);
                p[2 * st] = cadd(b2, b3);
                p[3 * st] = cmul(csub(b2, b3), w2);
            } else {
                const c64 t1 = cmulc(a1, w2), t3 = cmulc(a3, w2);
                const c64 b0 = cadd(a0, t1), b1 = csub(a0, t1), b2 = cadd(a2, t3), b3 = csub(a2, t3);
                const c64 u2 = cmulc(b2, w1), u3 = mul_pi(cmulc(b3, w1));
                p[0] = cadd(b0, u2);
                p[2 * st] = csub(b0, u2);
                p[st] = cadd(b1, u3);
                p[3 * st] = csub(b1, u3);
            }
        }
        __syncthreads();
    }
    if (!INV && (logL & 1)) {
        for (int t = threadIdx.x; t < nseq * (L >> 1); t += blockDim.x) {
            const int q = t / (L >> 1), u = t - q * (L >> 1);
            c64* p = buf + q * ss + (2 * u) * es;
            const c64 a = p[0], b = p[es];
            p[0] = cadd(a, b);
            p[es] = csub(a, b);
        }
        __syncthreads();
    }
}

// Column pass.  Forward: samples -> LDS [N1][C] -> transform over n1 -> times W_M^(n2 k1) -> work.  Inverse: work times
// the conjugate twiddle -> inverse transform -> samples (scaled by 1 / M) to `out`.  grid = (N2 / C), C columns each.
template <bool INV>
__global__ __launch_bounds__(256) void lp_cols_kernel(const float* __restrict__ in, c64* __restrict__ work_base,
                                                      double* __restrict__ out_base, LpBatch B, LpTables T) {
    extern __shared__ c64 lp_lds[];
    const LpSig sg = B.sigs ? B.sigs[blockIdx.y] : B.one;
    const LpGeom g = lp_geom(sg.lg);
    const int log1 = g.log1, log2 = g.log2, C = g.C;
    const int N1 = 1 << log1, N2 = 1 << log2;
    const int c0 = blockIdx.x * C;
    if (c0 >= N2) return;
    const float* x = in + sg.in_off;
    c64* work = work_base + sg.work_off;
    double* out = out_base + sg.out_off;
    const int64_t nx = sg.n;
    const double scale = 1.0 / (double)((int64_t)N1 << log2);
    for (int e = threadIdx.x; e < N1 * C; e += 256) {
        const int r = e / C, c = e - r * C, n2 = c0 + c;
        if (!INV) {
            const int64_t i0 = 2 * (((int64_t)r << log2) + n2) - ANTI_TURN_AROUND;
            c64 v;
            v.x = (i0 >= 0 && i0 < nx) ? (double)x[i0] : 0.0;
            v.y = (i0 + 1 >= 0 && i0 + 1 < nx) ? (double)x[i0 + 1] : 0.0;
            lp_lds[e] = v;
        } else {
            const int64_t k1 = bitrev(r, log1);
            lp_lds[e] = cmulc(work[((int64_t)r << log2) + n2], w_nfft(T, 2 * k1 * n2, sg.lg));
        }
    }
    __syncthreads();
    lds_fft<INV>(lp_lds, log1, C, C, 1, T.tw);
    for (int e = threadIdx.x; e < N1 * C; e += 256) {
        const int r = e / C, c = e - r * C, n2 = c0 + c;
        if (!INV) {
            const int64_t k1 = bitrev(r, log1);
            work[((int64_t)r << log2) + n2] = cmul(lp_lds[e], w_nfft(T, 2 * k1 * n2, sg.lg));
        } else {
            const int64_t i0 = 2 * (((int64_t)r << log2) + n2) - ANTI_TURN_AROUND;
            const c64 v = lp_lds[e];
            if (i0 >= 0 && i0 < nx) out[i0] = v.x * scale;
            if (i0 + 1 >= 0 && i0 + 1 < nx) out[i0 + 1] = v.y * scale;
        }
    }
}

// bins k and M - k of the real transform from Z[k], Z[M - k]; Praat's clearing of the packed array (1-based position
// 2k + 1 = real part, 2k + 2 = imaginary part of bin k, cleared from position `first_cleared`); back to Z'[k], Z'[M - k]
__device__ __forceinline__ void lp_filter_pair(c64& zk, c64& zm, int64_t k, int64_t M, int64_t first_cleared, const LpTables& T, int lg) {
    const c64 w = w_nfft(T, k, lg);
    const c64 E = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
    const c64 O = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));
    const c64 Tt = mul_pi(cmul(w, O));
    c64 xk = csub(E, Tt), xm = cconj(cadd(E, Tt));
    const int64_t m = M - k;
    if (2 * k + 1 >= first_cleared) xk.x = 0.0;
    if (2 * k + 2 >= first_cleared) xk.y = 0.0;
    if (2 * m + 1 >= first_cleared) xm.x = 0.0;
    if (2 * m + 2 >= first_cleared) xm.y = 0.0;
    const c64 s1 = cadd(xk, cconj(xm)), d1 = csub(xk, cconj(xm));
    const c64 s2 = cadd(xm, cconj(xk)), d2 = csub(xm, cconj(xk));
    const c64 r1 = mul_pi(cmulc(d1, w)), r2 = mul_pi(cmul(d2, w));
    zk = make_double2(0.5 * (s1.x + r1.x), 0.5 * (s1.y + r1.y));
    zm = make_double2(0.5 * (s2.x - r2.x), 0.5 * (s2.y - r2.y));
}

// Row pass: workgroup b owns the logical rows k1 = b and N1 - b (stored at their bit-reversed positions), b = 0 .. N1 / 2.
__global__ __launch_bounds__(256) void lp_rows_kernel(c64* __restrict__ work_base, LpBatch batch, LpTables T) {
    extern __shared__ c64 lp_lds[];
    const LpSig sg = batch.sigs ? batch.sigs[blockIdx.y] : batch.one;
    const LpGeom g = lp_geom(sg.lg);
    const int log1 = g.log1, log2 = g.log2, lg = sg.lg;
    const int N1 = 1 << log1, N2 = 1 << log2;
    if ((int)blockIdx.x > N1 / 2) return;
    c64* work = work_base + sg.work_off;
    const int64_t M = (int64_t)N1 << log2;
    const int64_t first_cleared = (int64_t)floor(batch.upfactor * (double)((int64_t)1 << lg));   // Praat: floor(upfactor * nfft)
    const int ka = blockIdx.x, kb = (N1 - ka) & (N1 - 1);
    const bool two = ka != kb;
    c64* rowA = work + ((int64_t)bitrev(ka, log1) << log2);
    c64* rowB = work + ((int64_t)bitrev(kb, log1) << log2);
    c64* A = lp_lds;
    c64* B = two ? lp_lds + N2 : lp_lds;
    for (int e = threadIdx.x; e < N2; e += 256) {
        A[e] = rowA[e];
        if (two) B[e] = rowB[e];
    }
    __syncthreads();
    lds_fft<false>(lp_lds, log2, two ? 2 : 1, 1, N2, T.tw);
    if (two) {
        for (int k2 = threadIdx.x; k2 < N2; k2 += 256) {
            const int pa = bitrev(k2, log2), pb = bitrev(N2 - 1 - k2, log2);
            c64 zk = A[pa], zm = B[pb];
            lp_filter_pair(zk, zm, ka + ((int64_t)k2 << log1), M, first_cleared, T, lg);
            A[pa] = zk;
            B[pb] = zm;
        }
    } else if (ka != 0) {                              // k1 = N1 / 2: the partner of k2 is N2 - 1 - k2 in the same row
        for (int k2 = threadIdx.x; k2 < N2 / 2; k2 += 256) {
            const int pa = bitrev(k2, log2), pb = bitrev(N2 - 1 - k2, log2);
            c64 zk = A[pa], zm = A[pb];
            lp_filter_pair(zk, zm, ka + ((int64_t)k2 << log1), M, first_cleared, T, lg);
            A[pa] = zk;
            A[pb] = zm;
        }
    } else {                                           // k1 = 0: partner N2 - k2; k2 = 0 holds DC and Nyquist, k2 = N2 / 2 is its own partner
        for (int k2 = threadIdx.x; k2 <= N2 / 2; k2 += 256) {
            if (k2 == 0) {
                const c64 z = A[0];
                const double dc = first_cleared > 1 ? z.x + z.y : 0.0;   // position 1; position 2 (Nyquist) is always cleared
                A[0] = make_double2(0.5 * dc, 0.5 * dc);
            } else {
                const int pa = bitrev(k2, log2), pb = bitrev(N2 - k2, log2);
                c64 zk = A[pa], zm = A[pb];
                lp_filter_pair(zk, zm, (int64_t)k2 << log1, M, first_cleared, T, lg);
                A[pa] = zk;
                if (pb != pa) A[pb] = zm;
            }
        }
    }
    __syncthreads();
    lds_fft<true>(lp_lds, log2, two ? 2 : 1, 1, N2, T.tw);
    for (int e = threadIdx.x; e < N2; e += 256) {
        rowA[e] = A[e];
        if (two) rowB[e] = B[e];
    }
}

// ---- NUM_interpolate_sinc on the re-centred grid -------------------------------------------------------------
// One thread per output sample.  SRC = double (the low-passed sound) or float (rate going up: no filter).
template <typename SRC>
__global__ __launch_bounds__(256) void praat_interp_kernel(const SRC* __restrict__ y, int64_t n_in, double fs_in, double fs_out,
                                                           int depth, float* __restrict__ out, int64_t n_out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const double dx_in = 1.0 / fs_in, dx_out = 1.0 / fs_out;
    const double duration = (double)n_in * dx_in;
    const double x1o = 0.5 * (duration - (double)(n_out - 1) / fs_out);
    const double x = (x1o + (double)o * dx_out - 0.5 * dx_in) / dx_in + 1.0;   // Praat's 1-based real index
    out[o] = (float)praat_interpolate_sinc(y, n_in, x, depth);
}

}  // namespace resample
}  // namespace rsaf

namespace rsaf {
namespace resample {

// device tables of the low-pass transform of nfft samples, cached per (device, nfft); angles reduced on the host
static int lp_tables(int64_t nfft, LpTables* out) {
    static std::mutex mu;
    static std::map<std::pair<int, int64_t>, c64*> cache;
    int dev = 0;
    RSAF_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto get = [&](int64_t key, int64_t count, int64_t num_step, int64_t den, c64** p) -> int {
        auto it = cache.find({dev, key});
        if (it == cache.end()) {
            std::vector<c64> h((size_t)count);
            for (int64_t j = 0; j < count; ++j) {
                const long double a = 2.0L * 3.141592653589793238462643383279502884L * (long double)(j * num_step) / (long double)den;
                h[(size_t)j] = make_double2((double)cosl(a), (double)-sinl(a));
            }
            c64* d = nullptr;
            RSAF_CHECK_HIP(hipMalloc(&d, (size_t)count * sizeof(c64)));
            RSAF_CHECK_HIP(hipMemcpy(d, h.data(), (size_t)count * sizeof(c64), hipMemcpyHostToDevice));
            it = cache.emplace(std::make_pair(dev, key), d).first;
        }
        *p = it->second;
        return RSAF_OK;
    };
    c64 *tw = nullptr, *lo = nullptr, *hi = nullptr;
    int rc = get(-1, TW_N, 1, TW_N, &tw);                                        // key -1: the butterfly table
    if (rc != RSAF_OK) return rc;
    rc = get(2 * nfft, TW_N, 1, nfft, &lo);
    if (rc != RSAF_OK) return rc;
    const int64_t nhi = nfft > TW_N ? nfft / TW_N : 1;
    rc = get(2 * nfft + 1, nhi, TW_N, nfft, &hi);
    if (rc != RSAF_OK) return rc;
    out->tw = tw; out->lo = lo; out->hi = hi; out->lg_max = 0;
    return RSAF_OK;
}

}  // namespace resample
}  // namespace rsaf

using namespace rsaf;

extern "C" {

int rsaf_resample_sinc_hann(const float* in, int64_t n_in, const float* taps, const int* tap_start, int n_phase, int orig,
                            int taps_per_phase, float* out, int64_t n_out, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_in >= 0 && n_out >= 0 && n_phase >= 1 && orig >= 1 && taps_per_phase >= 1, "bad sizes");
    if (n_out == 0) return RSAF_OK;
    RSAF_CHECK_ARG(in && taps && tap_start && out, "NULL pointer");
    RSAF_CHECK_ARG((n_out + 255) / 256 <= 0x7fffffffLL, "output too long for one launch");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("resample_sinc_hann", s, 0.0, 4.0 * (double)(n_in + n_out));
    hipLaunchKernelGGL(resample::sinc_hann_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, in, n_in, taps,
                       tap_start, n_phase, orig, taps_per_phase, out, n_out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int64_t rsaf_resample_praat_work_bytes(int64_t n_in, double fs_in, double fs_out) {
    if (n_in <= 0 || !(fs_in > 0.0) || !(fs_out > 0.0) || fs_out * (1.0 / fs_in) >= 1.0) return 0;
    int64_t nfft = 1;
    while (nfft < n_in + 2 * resample::ANTI_TURN_AROUND) nfft *= 2;
    return nfft * 8 + n_in * 8;                         // nfft / 2 complex numbers + the low-passed sound (fp64)
}

}  // extern "C"

// the three passes of the low-pass over a batch (sigs on the device, n_sigs of them) or over `one` sound
static int launch_lowpass(const float* in, double* out, resample::c64* work, const resample::LpSig* sigs, int n_sigs,
                          const resample::LpSig& one, int lg_max, double upfactor, hipStream_t s) {
    using namespace resample;
    RSAF_CHECK_ARG(lg_max >= 11 && lg_max <= 24, "sound longer than 2^24 - 2000 samples: the low-pass transform does not fit its two LDS passes");
    LpTables T;
    {
        const int rc = lp_tables((int64_t)1 << lg_max, &T);
        if (rc != RSAF_OK) return rc;
    }
    T.lg_max = lg_max;
    unsigned gc = 1, gr = 1;
    size_t lds_c = 0, lds_r = 0;
    for (int lg = 11; lg <= lg_max; ++lg) {             // the batch may hold any shorter transform
        const LpGeom g = lp_geom(lg);
        gc = std::max(gc, (unsigned)((1 << g.log2) / g.C));
        gr = std::max(gr, (unsigned)((1 << g.log1) / 2 + 1));
        lds_c = std::max(lds_c, ((size_t)g.C << g.log1) * sizeof(c64));
        lds_r = std::max(lds_r, ((size_t)2 << g.log2) * sizeof(c64));
    }
    if (lds_r > 48 * 1024)
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)lp_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
    if (lds_c > 48 * 1024) {
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)lp_cols_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)lp_cols_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
    }
    LpBatch B;
    B.sigs = sigs;
    B.one = one;
    B.upfactor = upfactor;
    const unsigned ny = sigs ? (unsigned)n_sigs : 1u;
    hipLaunchKernelGGL(lp_cols_kernel<false>, dim3(gc, ny), dim3(256), lds_c, s, in, work, (double*)nullptr, B, T);
    hipLaunchKernelGGL(lp_rows_kernel, dim3(gr, ny), dim3(256), lds_r, s, work, B, T);
    hipLaunchKernelGGL(lp_cols_kernel<true>, dim3(gc, ny), dim3(256), lds_c, s, (const float*)nullptr, work, out, B, T);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

extern "C" {

int rsaf_praat_lowpass_batch(const float* in, const void* sigs, int n_sigs, int lg_max, double upfactor, void* work,
                             int64_t work_complex, double* out, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_sigs >= 0 && n_sigs <= 65535 && upfactor > 0.0 && upfactor < 1.0, "bad argument");
    if (n_sigs == 0) return RSAF_OK;
    RSAF_CHECK_ARG(in && sigs && work && out && work_complex >= 1, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("praat_lowpass_fft", s, 0.0, 6.0 * 16.0 * (double)work_complex);
    resample::LpSig none{};
    return launch_lowpass(in, out, (resample::c64*)work, (const resample::LpSig*)sigs, n_sigs, none, lg_max, upfactor, s);
}

int rsaf_resample_praat(const float* in, int64_t n_in, double fs_in, double fs_out, int precision, float* out,
                        int64_t n_out, void* work, int64_t work_bytes, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_in >= 0 && n_out >= 0 && fs_in > 0.0 && fs_out > 0.0 && precision >= 1 && precision <= 4096, "bad arguments");
    if (n_out == 0) return RSAF_OK;
    RSAF_CHECK_ARG(in && out && n_in >= 1, "NULL pointer or empty input");
    RSAF_CHECK_ARG((n_out + 255) / 256 <= 0x7fffffffLL, "output too long for one launch");
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n_out + 255) / 256);
    const double upfactor = fs_out * (1.0 / fs_in);     // Praat: samplingFrequency * my dx
    if (!(upfactor < 1.0)) {                            // rate going up: interpolation only
        ProfScope prof("resample_praat", s, 0.0, 4.0 * (double)(n_in + n_out));
        hipLaunchKernelGGL(resample::praat_interp_kernel<float>, dim3(grid), dim3(256), 0, s, in, n_in, fs_in, fs_out, precision,
                           out, n_out);
        RSAF_CHECK_HIP(hipGetLastError());
        return RSAF_OK;
    }
    int64_t nfft = 1;
    int lg = 0;
    while (nfft < n_in + 2 * resample::ANTI_TURN_AROUND) { nfft *= 2; ++lg; }
    RSAF_CHECK_ARG(lg <= 24, "sound longer than 2^24 - 2000 samples: the low-pass transform does not fit its two LDS passes");
    RSAF_CHECK_ARG(work && work_bytes >= rsaf_resample_praat_work_bytes(n_in, fs_in, fs_out), "workspace missing or too small");
    resample::c64* wk = (resample::c64*)work;
    double* lp = (double*)(wk + nfft / 2);
    ProfScope prof("resample_praat", s, 2.0 * 2.5 * (double)nfft * (double)(lg - 1), 6.0 * 8.0 * (double)nfft + 4.0 * (double)(n_in + n_out));
    resample::LpSig one{};
    one.n = (int)n_in;
    one.lg = lg;
    const int rc = launch_lowpass(in, lp, wk, nullptr, 1, one, lg, upfactor, s);
    if (rc != RSAF_OK) return rc;
    hipLaunchKernelGGL(resample::praat_interp_kernel<double>, dim3(grid), dim3(256), 0, s, (const double*)lp, n_in, fs_in, fs_out,
                       precision, out, n_out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

}  // extern "C"

// ---- PCM decode + mono mix-down (the first step of SURVEY.md §8f rank 1) --------------------------------------
// Interleaved little-endian integer PCM of 1, 2, 3 or 4 bytes per sample -> float32 in [-1, 1) per channel
// (8-bit is unsigned, as in WAV), then the channel mean in channel order, float32 arithmetic: the same numbers
// as torchaudio.load(...).mean(dim=0) (src/foundation_model_extractor.py:87-91) and wavio.read_wav_mono.
namespace rsaf {
namespace resample {

__global__ __launch_bounds__(256) void pcm_to_mono_kernel(const unsigned char* __restrict__ pcm, int width, int n_ch,
                                                          int64_t n_frames, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_frames) return;
    const unsigned char* p = pcm + i * n_ch * width;
    float acc = 0.0f;
    for (int c = 0; c < n_ch; ++c, p += width) {
        float v;
        if (width == 2) {
            v = (float)(short)(p[0] | (p[1] << 8)) / 32768.0f;
        } else if (width == 1) {
            v = ((float)p[0] - 128.0f) / 128.0f;
        } else if (width == 3) {
            int s = p[0] | (p[1] << 8) | (p[2] << 16);
            s = s >= (1 << 23) ? s - (1 << 24) : s;
            v = (float)((double)s / 8388608.0);
        } else {
            const int s = (int)((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24));
            v = (float)((double)s / 2147483648.0);
        }
        acc = c == 0 ? v : acc + v;
    }
    out[i] = n_ch > 1 ? acc / (float)n_ch : acc;
}

}  // namespace resample
}  // namespace rsaf

extern "C" int rsaf_pcm_to_mono_f32(const void* pcm, int sample_width, int n_channels, int64_t n_frames, float* out,
                                    rsaf_stream_t stream) {
    RSAF_CHECK_ARG(sample_width >= 1 && sample_width <= 4 && n_channels >= 1 && n_frames >= 0, "bad PCM geometry");
    if (n_frames == 0) return RSAF_OK;
    RSAF_CHECK_ARG(pcm && out, "NULL pointer");
    RSAF_CHECK_ARG((n_frames + 255) / 256 <= 0x7fffffffLL, "too many frames for one launch");
    hipStream_t s = (hipStream_t)stream;
    rsaf::ProfScope prof("pcm_to_mono", s, 0.0, (double)n_frames * (n_channels * sample_width + 4));
    hipLaunchKernelGGL(rsaf::resample::pcm_to_mono_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, s,
                       (const unsigned char*)pcm, sample_width, n_channels, n_frames, out);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}
