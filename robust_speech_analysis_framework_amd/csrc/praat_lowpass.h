// Praat's whole-sound FFT low-pass (Sound_resample, anti-aliasing branch) for gfx950: device code shared by the resampler in
// front of the extractors (resample.hip), the 10 kHz resampling of To Formant (burg) and the per-interval one of To
// PowerCepstrogram (mshds_cpp.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rsaf {
namespace resample {

// ---- whole-sound FFT low-pass (Praat Sound_resample, anti-aliasing branch) -------------------------------------
// The nfft real samples (1 000 zeros, the sound, zeros) are the M = nfft / 2 complex numbers z[n] = (d[2n], d[2n+1]),
// n = n1 N2 + n2.  Forward: column transforms over n1 (decimation in frequency, in place in LDS, output row r holds
// k1 = bitrev(r)), times W_M^(n2 k1); then row transforms over n2 (same scheme, LDS position p holds k2 = bitrev(p)):
// Z[k1 + N1 k2].  The real-transform bins X[k], X[M - k] come from Z[k], Z[M - k], which live in rows k1 and N1 - k1: one
// workgroup owns both rows, clears what Praat clears, folds back to Z' and runs the inverse row transforms (decimation
// in time: bit-reversed in, natural out).  The inverse column pass undoes the first one.  No pass reorders memory.
typedef double2 c64;
constexpr int TW_LOG = 13, TW_N = 1 << TW_LOG;        // W_8192^j: butterflies of every LDS transform (length <= 8192)
constexpr int LP_LG_MAX = 26;                         // longest transform: 2^26 samples (25 min at 44.1 kHz, 69 min at 16 kHz)
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
    const c64* tw;      // [TW_N]  e^(-2 pi i j / TW_N)
    const c64* lo;      // [TW_N]  e^(-2 pi i j / nfft_max)
    const c64* hi;      // [nfft_max / TW_N or 1]  e^(-2 pi i TW_N j / nfft_max)
    int lg_max;         // the tables belong to nfft_max = 2^lg_max; a shorter transform strides through them
};

// one sound of a batch: n samples at in + in_off -> out + out_off, transform of 2^lg samples in work + work_off
struct LpSig {
    int64_t in_off, out_off, work_off;
    int n, lg;
};
static_assert(sizeof(LpSig) == 32, "LpSig layout");
// what the row pass does to the spectrum between the forward and the inverse transform
enum LpMode {
    LP_LOWPASS = 0,         // Sound_resample, anti-aliasing branch: clear the packed array from floor(upfactor * nfft)
    LP_UPSAMPLE_EVEN = 1,   // Sound_upsample: linear ramp to zero over the last 5 % of the packed array; the even output samples
    LP_UPSAMPLE_ODD = 2     // ... and the odd ones: bin k also turned by e^(+i pi k / nfft) (half an input sample later)
};
struct LpBatch {
    const LpSig* sigs;  // device array indexed by the workgroup's y index, or nullptr: `one`
    LpSig one;
    double upfactor;
    int mode;           // LpMode
};

struct LpGeom { int log1, log2, C, logC; };
__host__ __device__ inline LpGeom lp_geom(int lg) {
    LpGeom g;
    const int logM = lg - 1;                            // lg >= 11
    // about square; rows of at most 2 048 points (two rows = 64 KB of LDS), columns of at most 4 096 (64 KB); the two
    // longest transforms (2^25, 2^26 samples) take rows of 4 096 and columns of up to 8 192 points: 128 KB of the 160 KB
    const int row_cap = lg <= 24 ? 11 : 12, col_cap = lg <= 24 ? 4096 : 8192;
    g.log2 = (logM + 1) / 2 < row_cap ? (logM + 1) / 2 : row_cap;
    g.log1 = logM - g.log2;
    g.C = 8;
    g.logC = 3;
    while (g.C > 1 && ((int64_t)g.C << g.log1) > col_cap) { g.C >>= 1; --g.logC; }
    return g;
}

// e^(-2 pi i p / 2^lg), 0 <= p < 2^lg
__device__ __forceinline__ c64 w_nfft(const LpTables& T, int64_t p, int lg) {
    p <<= T.lg_max - lg;
    return cmul(T.lo[p & (TW_N - 1)], T.hi[p >> TW_LOG]);
}

// 2^lognseq interleaved sequences of length 2^logL in LDS (element e of sequence q at buf[e * es + q * ss]); INV = false:
// decimation in frequency, forward twiddles, natural in / bit-reversed out; INV = true: decimation in time, conjugate
// twiddles, bit-reversed in / natural out, unnormalised.  Two butterfly layers per barrier.  Consecutive threads take
// consecutive sequences when those lie closer together than the elements of one sequence (column transforms), else
// consecutive butterflies of one sequence: neighbouring lanes stay 16 bytes apart either way.
template <bool INV>
__device__ inline void lds_fft(c64* buf, int logL, int lognseq, int es, int ss, const c64* __restrict__ tw) {
    const bool seq_fast = ss < es;
    const int qmask = (1 << lognseq) - 1;
    auto radix2 = [&]() {
        const int lc = logL - 1;                                      // log2 of the butterflies per sequence
        for (int t = threadIdx.x; t < (1 << (lognseq + lc)); t += blockDim.x) {
            const int q = seq_fast ? (t & qmask) : (t >> lc), u = seq_fast ? (t >> lognseq) : (t & ((1 << lc) - 1));
            c64* p = buf + q * ss + (2 * u) * es;
            const c64 a = p[0], b = p[es];
            p[0] = cadd(a, b);
            p[es] = csub(a, b);
        }
        __syncthreads();
    };
    if (INV && (logL & 1)) radix2();
    const int first = INV ? (2 + (logL & 1)) : logL, last = INV ? logL : (2 + (logL & 1));
    for (int sl = first; INV ? sl <= last : sl >= last; sl += INV ? 2 : -2) {
        const int ql = sl - 2, qn = 1 << ql;
        const int lc = logL - 2;
        for (int t = threadIdx.x; t < (1 << (lognseq + lc)); t += blockDim.x) {
            const int q = seq_fast ? (t & qmask) : (t >> lc), u = seq_fast ? (t >> lognseq) : (t & ((1 << lc) - 1));
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
    if (!INV && (logL & 1)) radix2();
}

// Column pass.  Forward: samples -> LDS [N1][C] -> transform over n1 -> times W_M^(n2 k1) -> work.  Inverse: work times
// the conjugate twiddle -> inverse transform -> samples (scaled by 1 / M) to `out`.  grid = (N2 / C), C columns each.
// (bx = block index within the sound; every thread of the workgroup takes the same path)
template <bool INV>
__device__ void lp_cols_body(const float* __restrict__ in, c64* __restrict__ work_base, double* __restrict__ out_base,
                             const LpSig& sg, int bx, const LpTables& T, c64* lp_lds) {
    const LpGeom g = lp_geom(sg.lg);
    const int log1 = g.log1, log2 = g.log2, C = g.C;
    const int N1 = 1 << log1, N2 = 1 << log2;
    const int c0 = bx * C;
    if (c0 >= N2) return;
    const float* x = in + sg.in_off;
    c64* work = work_base + sg.work_off;
    double* out = out_base + sg.out_off;
    const int64_t nx = sg.n;
    const double scale = 1.0 / (double)((int64_t)N1 << log2);
    for (int e = threadIdx.x; e < N1 * C; e += blockDim.x) {
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
    lds_fft<INV>(lp_lds, log1, g.logC, C, 1, T.tw);
    for (int e = threadIdx.x; e < N1 * C; e += blockDim.x) {
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
// Sound_upsample (mode != LP_LOWPASS): instead of the clearing, the 1-based packed positions i > imin = (integer)(0.95 nfft)
// are scaled by (nfft - i) / (nfft - imin) (real and imaginary part of a bin sit at different positions and get different
// factors), the Nyquist position is cleared, and the array is transformed back at TWICE the length: output sample 2p is the
// inverse transform of this spectrum at p, output sample 2p + 1 that of the spectrum with bin k turned by e^(+i pi k / nfft)
// (the tables must then reach 2 nfft: T.lg_max >= lg + 1).
__device__ __forceinline__ void lp_filter_pair(c64& zk, c64& zm, int64_t k, int64_t M, int64_t first_cleared, const LpTables& T, int lg,
                                               int mode = LP_LOWPASS) {
    const c64 w = w_nfft(T, k, lg);
    const c64 E = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
    const c64 O = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));
    const c64 Tt = mul_pi(cmul(w, O));
    c64 xk = csub(E, Tt), xm = cconj(cadd(E, Tt));
    const int64_t m = M - k;
    if (mode == LP_LOWPASS) {
        if (2 * k + 1 >= first_cleared) xk.x = 0.0;
        if (2 * k + 2 >= first_cleared) xk.y = 0.0;
        if (2 * m + 1 >= first_cleared) xm.x = 0.0;
        if (2 * m + 2 >= first_cleared) xm.y = 0.0;
    } else {
        const int64_t nfft = 2 * M, imin = (int64_t)((double)nfft * 0.95);
        auto ramp = [&](int64_t pos) -> double { return pos > imin ? (double)(nfft - pos) / (double)(nfft - imin) : 1.0; };
        xk.x *= ramp(2 * k + 1); xk.y *= ramp(2 * k + 2);
        xm.x *= ramp(2 * m + 1); xm.y *= ramp(2 * m + 2);
        if (mode == LP_UPSAMPLE_ODD) {
            xk = cmulc(xk, w_nfft(T, k, lg + 1));
            xm = cmulc(xm, w_nfft(T, m, lg + 1));
        }
    }
    const c64 s1 = cadd(xk, cconj(xm)), d1 = csub(xk, cconj(xm));
    const c64 s2 = cadd(xm, cconj(xk)), d2 = csub(xm, cconj(xk));
    const c64 r1 = mul_pi(cmulc(d1, w)), r2 = mul_pi(cmul(d2, w));
    zk = make_double2(0.5 * (s1.x + r1.x), 0.5 * (s1.y + r1.y));
    zm = make_double2(0.5 * (s2.x - r2.x), 0.5 * (s2.y - r2.y));
}

// the filter step for element k2 of the logical row ka (A) and its partner in row N1 - ka (B; B == A when the row is its own
// partner); both rows hold their transform in bit-reversed positions
__device__ __forceinline__ void lp_filter_at(c64* A, c64* B, int ka, int k2, int log1, int log2, int64_t first_cleared,
                                             const LpTables& T, int lg, int mode = LP_LOWPASS) {
    const int N1 = 1 << log1, N2 = 1 << log2;
    const int64_t M = (int64_t)N1 << log2;
    const bool two = ka != ((N1 - ka) & (N1 - 1));
    if (two) {
        const int pa = bitrev(k2, log2), pb = bitrev(N2 - 1 - k2, log2);
        c64 zk = A[pa], zm = B[pb];
        lp_filter_pair(zk, zm, ka + ((int64_t)k2 << log1), M, first_cleared, T, lg, mode);
        A[pa] = zk;
        B[pb] = zm;
    } else if (ka != 0) {                              // k1 = N1 / 2: the partner of k2 is N2 - 1 - k2 in the same row
        if (k2 >= N2 / 2) return;
        const int pa = bitrev(k2, log2), pb = bitrev(N2 - 1 - k2, log2);
        c64 zk = A[pa], zm = A[pb];
        lp_filter_pair(zk, zm, ka + ((int64_t)k2 << log1), M, first_cleared, T, lg, mode);
        A[pa] = zk;
        A[pb] = zm;
    } else {                                           // k1 = 0: partner N2 - k2; k2 = 0 holds DC and Nyquist, k2 = N2 / 2 is its own partner
        if (k2 > N2 / 2) return;
        if (k2 == 0) {
            const c64 z = A[0];
            const double dc = (mode != LP_LOWPASS || first_cleared > 1) ? z.x + z.y : 0.0;   // position 1; position 2 (Nyquist) is always cleared
            A[0] = make_double2(0.5 * dc, 0.5 * dc);
        } else {
            const int pa = bitrev(k2, log2), pb = bitrev(N2 - k2, log2);
            c64 zk = A[pa], zm = A[pb];
            lp_filter_pair(zk, zm, (int64_t)k2 << log1, M, first_cleared, T, lg, mode);
            A[pa] = zk;
            if (pb != pa) A[pb] = zm;
        }
    }
}

// Row pass: workgroup b owns the logical rows k1 = b and N1 - b (stored at their bit-reversed positions), b = 0 .. N1 / 2.
__device__ inline void lp_rows_body(c64* __restrict__ work_base, const LpSig& sg, int bx, double upfactor, const LpTables& T,
                                    c64* lp_lds, int mode = LP_LOWPASS) {
    const LpGeom g = lp_geom(sg.lg);
    const int log1 = g.log1, log2 = g.log2, lg = sg.lg;
    const int N1 = 1 << log1, N2 = 1 << log2;
    if (bx > N1 / 2) return;
    c64* work = work_base + sg.work_off;
    const int64_t first_cleared = (int64_t)floor(upfactor * (double)((int64_t)1 << lg));   // Praat: floor(upfactor * nfft)
    const int ka = bx, kb = (N1 - ka) & (N1 - 1);
    const bool two = ka != kb;
    c64* rowA = work + ((int64_t)bitrev(ka, log1) << log2);
    c64* rowB = work + ((int64_t)bitrev(kb, log1) << log2);
    c64* A = lp_lds;
    c64* B = two ? lp_lds + N2 : lp_lds;
    for (int e = threadIdx.x; e < N2; e += blockDim.x) {
        A[e] = rowA[e];
        if (two) B[e] = rowB[e];
    }
    __syncthreads();
    lds_fft<false>(lp_lds, log2, two ? 1 : 0, 1, N2, T.tw);
    for (int k2 = threadIdx.x; k2 < N2; k2 += blockDim.x) lp_filter_at(A, B, ka, k2, log1, log2, first_cleared, T, lg, mode);
    __syncthreads();
    lds_fft<true>(lp_lds, log2, two ? 1 : 0, 1, N2, T.tw);
    for (int e = threadIdx.x; e < N2; e += blockDim.x) {
        rowA[e] = A[e];
        if (two) rowB[e] = B[e];
    }
}

// The whole low-pass of one short sound inside LDS (2^(lg-1) complex numbers: 64 KB at lg = 13, 128 KB at lg = 14): the same
// column / row / filter / row / column sequence without the trips through the work buffer.  Called by every thread of the
// workgroup; x = the sound's samples, out = its low-passed samples.
__device__ inline void lp_whole_in_lds(const float* __restrict__ x, double* __restrict__ out, int n, int lg, double upfactor,
                                       const LpTables& T, c64* buf) {
    const LpGeom g = lp_geom(lg);
    const int log1 = g.log1, log2 = g.log2;
    const int N1 = 1 << log1, N2 = 1 << log2, M = N1 << log2;
    const int64_t first_cleared = (int64_t)floor(upfactor * (double)((int64_t)1 << lg));
    for (int e = threadIdx.x; e < M; e += blockDim.x) {
        const int i0 = 2 * e - ANTI_TURN_AROUND;
        c64 v;
        v.x = (i0 >= 0 && i0 < n) ? (double)x[i0] : 0.0;
        v.y = (i0 + 1 >= 0 && i0 + 1 < n) ? (double)x[i0 + 1] : 0.0;
        buf[e] = v;
    }
    __syncthreads();
    lds_fft<false>(buf, log1, log2, N2, 1, T.tw);                      // columns: element r of column c at buf[r N2 + c]
    for (int e = threadIdx.x; e < M; e += blockDim.x) {
        const int r = e >> log2, n2 = e & (N2 - 1);
        buf[e] = cmul(buf[e], w_nfft(T, 2 * (int64_t)bitrev(r, log1) * n2, lg));
    }
    __syncthreads();
    lds_fft<false>(buf, log2, log1, 1, N2, T.tw);                      // rows
    for (int e = threadIdx.x; e < (N1 / 2 + 1) * N2; e += blockDim.x) {
        const int ka = e >> log2, k2 = e & (N2 - 1);
        const int kb = (N1 - ka) & (N1 - 1);
        lp_filter_at(buf + ((int64_t)bitrev(ka, log1) << log2), buf + ((int64_t)bitrev(kb, log1) << log2), ka, k2, log1, log2,
                     first_cleared, T, lg);
    }
    __syncthreads();
    lds_fft<true>(buf, log2, log1, 1, N2, T.tw);
    for (int e = threadIdx.x; e < M; e += blockDim.x) {
        const int r = e >> log2, n2 = e & (N2 - 1);
        buf[e] = cmulc(buf[e], w_nfft(T, 2 * (int64_t)bitrev(r, log1) * n2, lg));
    }
    __syncthreads();
    lds_fft<true>(buf, log1, log2, N2, 1, T.tw);
    const double scale = 1.0 / (double)M;
    for (int e = threadIdx.x; e < M; e += blockDim.x) {
        const int i0 = 2 * e - ANTI_TURN_AROUND;
        const c64 v = buf[e];
        if (i0 >= 0 && i0 < n) out[i0] = v.x * scale;
        if (i0 + 1 >= 0 && i0 + 1 < n) out[i0 + 1] = v.y * scale;
    }
    __syncthreads();
}

// device tables of the transform of nfft samples, cached per (device, nfft) (resample.hip); the caller sets lg_max
int lp_tables(int64_t nfft, LpTables* out);

}  // namespace resample
}  // namespace rsaf
