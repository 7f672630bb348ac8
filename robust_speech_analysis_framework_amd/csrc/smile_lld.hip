// openSMILE-style low-level descriptors for gfx950 (wave64), at the file's own sample rate, in FLOAT64.
//
// One fused kernel for Androids.conf:73-186 and :258-280 of the reference
// (cFramer -> cVectorPreemphasis -> cWindower -> cTransformFFT -> cFFTmagphase ->
//  {cMelspec -> cMfcc, cEnergy, cMZcr, cIntensity, cSpectral, cSpecScale -> cPitchShs}); the reference runs that
// chain by spawning SMILExtract once per file (src/opensmile_extractor.py:62-75).  The sequential tail of the
// pitch chain (cPitchSmootherViterbi, cValbasedSelector, cPitchJitter) is csrc/smile_pitch.hip.
//
// Why float64.  The chain is full of DECISIONS: the local-maximum test of the peak enhancement on every bin (also the
// noise-floor bins, whose float32 FFT error is of the order of the bins themselves), peak picking and ranking of the
// sub-harmonic summation, the Viterbi path, the lags of the jitter search, the roll-off threshold crossings, and the
// first-occurrence positions (maxPos / minPos) of all 76 contours.  north_star asks for positions bit-exact and values
// within 1e-4 of the CPU path; the CPU restatement (oracle/smile_oracle.py) is float64.  Rounds 1-2 ran this kernel in
// packed float32: 97-99.5 % of the frames took the oracle's decisions and the rest did not.  In float64 the two sides
// differ by ~1e-15 and a decision flips only on a tie of that size.  The stage is < 1 % of the hot path's time, so
// the precision is spent here.
//
// Mapping.  One wave per run of RUNW consecutive frames of one clip (64-thread workgroups: no workgroup-level
// synchronisation anywhere).  Per frame: samples -> pre-emphasis / Hamming in registers -> the NFFT-point real FFT as
// an NFFT/2-point complex Stockham radix-4 (+ one radix-2 stage when log2 is odd) in the wave's LDS buffer, in place
// (a wave is lock-step: every lane reads its butterflies before any lane writes) -> magnitudes (registers + LDS).
// Wave-wide sums go 8 at a time through an LDS transpose (8 writes, 8 reads, 3 lane exchanges instead of 8 x 6
// dependent exchanges).  The tridiagonal system of the natural spline and the cumulative sums of the roll-off points
// are lane-blocked scans of affine maps / partial sums (exact to rounding, no truncation).  Everything that is a
// scalar function of a frame's sums (sqrt, pow, log, exp, the moment ratios) is deferred to the end of the run, where
// lane t finishes frame t: the transcendental code runs once per run instead of once per frame.
// Spectral flux needs the previous frame's magnitudes: the wave transforms the frame in front of its run once more
// (window + FFT only), which keeps the waves independent of each other.
//
// Templated on the FFT length (256 / 512 / 1024 / 2048 <-> 8 / 16 / 22.05-32 / 44.1-48 kHz; frame and hop lengths are
// run-time).  Semantics are those of oracle/smile_oracle.py (the CPU restatement), parity unpinned.
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "rsaf_common.h"

namespace rsaf {
namespace smile {

typedef double2 c64;

constexpr int NLLD = RSAF_SMILE_NLLD;
constexpr int NMEL = 26;
constexpr int NMFCC = 12;
constexpr int NCAND = RSAF_SMILE_NCAND;
constexpr int NHARM = 15;
constexpr double PREEMPH = 0.97;
constexpr double HTK_SCALE = 32767.0;
constexpr double MEL_FLOOR = 1.0;
constexpr double I0 = 1e-6;
constexpr double SHS_MINPITCH = 52.0, SHS_MAXPITCH = 620.0;
constexpr double LN2 = 0.69314718055994530942;

constexpr int RUNW = 16;                 // frames per wave
constexpr int NSUM = 23;                 // doubles a frame leaves for the end-of-run pass (slots 0 .. 22)
constexpr int RED_D = 8 * 65;            // doubles of the 8-sums-at-a-time reduction scratch

template <int LOG2N>
struct Geo {
    static constexpr int NFFT = 1 << LOG2N;
    static constexpr int NC = NFFT / 2;              // complex points of the packed-real transform
    static constexpr int NB = NC + 1;                // magnitude bins
    static constexpr int PPL = NC / 64;              // bins per lane (2, 4, 8, 16)
    static constexpr int NBP = NC + 8;               // padded length of the per-bin tables
    // LDS of one wave, in doubles.  Round 4: the octave spectrum no longer has a buffer of its own - it lives in the second half
    // of the dead FFT buffer plus a zero pad behind it, the spline moments take the first half once the enhanced spectrum is
    // consumed, and the two magnitude arrays swap roles from frame to frame (this frame's magnitudes ARE the next frame's
    // "previous" ones: no copy; the smoothed and the summation spectrum go to the array the flux has finished with):
    // 15.6 -> 12.7 KB per wave at 16 kHz, i.e. 12 instead of 10 waves per CU.
    // The 2 048-point instance (44.1 / 48 kHz) is trimmed to the last double: 40 928 bytes per wave = FOUR waves per CU instead
    // of three (its arrays need NC + 1 entries and the FFT buffer exactly 2 NC; the shorter instances keep the padding the
    // reduction scratch in the FFT buffer relies on).
    static constexpr bool TRIM = LOG2N >= 11;
    static constexpr int Z_D = 2 * NC + (TRIM ? 0 : 8);   // FFT buffer (c64[NC]); later the enhanced spectrum / spline moments | octave spectrum
    static constexpr int ARR = NC + (TRIM ? 2 : 8);       // one per-bin array (NB = NC + 1 entries)
    static constexpr int PAD_D = ((NC * 5) / 8 + 8 + 7) & ~7;   // zeros behind the octave spectrum: longer than the largest harmonic shift (< 0.62 NC)
    static constexpr int S_D = (Z_D - ARR) + PAD_D;  // the octave spectrum's room: NB bins + the pad
    static_assert(S_D >= NB + (NC * 5) / 8 + 1, "octave spectrum + pad must fit");
    static constexpr int MELCAP = NC / 4;            // cap of the longest side of a triangular mel band, in bins
    static constexpr bool RED_IN_Z = Z_D >= RED_D;   // the reduction scratch lives in the FFT buffer when it fits
    static constexpr int OFF_S = ARR, OFF_P0 = Z_D + PAD_D, OFF_P1 = OFF_P0 + ARR, OFF_STASH = OFF_P1 + ARR;
    static constexpr int OFF_RED = RED_IN_Z ? 0 : OFF_STASH + RUNW * NSUM;
    static constexpr int WAVE_D = OFF_STASH + RUNW * NSUM + (RED_IN_Z ? 0 : RED_D);
};

// ---- constant tables (host-computed in double, one blob per (device, sample rate)) -------------------
template <int LOG2N>
struct __attribute__((aligned(16))) Tables {
    using G = Geo<LOG2N>;
    c64 tw[G::NC + G::NC / 2];        // exp(-2 pi i k / NFFT), k < 3 NC / 2: the packed-real unpack uses k <= NC, the
                                      // complex stages exp(-2 pi i m / NC) = tw[2 m], m < 3 NC / 4
    double ham[G::NFFT];              // Hamming window, zero beyond the frame
    double lo_wt[G::NBP];             // HTK lower-channel weight per bin (0 where unused)
    double sharp[G::NBP];             // bark(f) * g(bark) per bin
    double tb[G::NBP];                // octave-scale target i: fractional position inside source interval klo[i]
    double audw[G::NBP];              // auditory weighting of target i
    double sp_g[G::NBP];              // tridiagonal (1, 4, 1) elimination factors, 0 at bin 0
    int klo[G::NBP];
    double dct[NMEL * 16];            // DCT-II rows 1..12 with the lifter folded in, transposed: [band j][coefficient k] (16 per row)
    int seg_start[32];                // bins with lower channel c are [seg_start[c], seg_start[c+1]), c = 0..26
    double mel_w[G::MELCAP * 64];     // lane 2 c + side = one side of triangular band c: weight of its i-th bin at [i * 64 + lane]
    int mel_b0[64];                   // ... first bin of that side
    int mel_len, mel_pad[3];          // bins of the longest side
    int shs_shift[16];
    double shs_w[16];
    double ham_sum, df, fmin_l2, dl2;
    double slope_sf, slope_den;
    int frame, hop, fs, pad;
};

static double mel_d(double f) { return 2595.0 * std::log10(1.0 + f / 700.0); }

template <int LOG2N>
static void build_tables(Tables<LOG2N>& t, int fs, int frame, int hop) {
    using G = Geo<LOG2N>;
    std::memset(&t, 0, sizeof(t));
    t.fs = fs; t.frame = frame; t.hop = hop;
    const double df = (double)fs / G::NFFT;
    t.df = df;
    double hs = 0;
    for (int i = 0; i < frame; ++i) {
        const double w = 0.54 - 0.46 * std::cos(2.0 * M_PI * i / (frame - 1));
        t.ham[i] = w;
        hs += w;
    }
    t.ham_sum = hs;
    for (int m = 0; m < G::NC + G::NC / 2; ++m) {
        const long double a = 2.0L * 3.141592653589793238462643383279502884L * m / G::NFFT;
        t.tw[m] = make_double2((double)cosl(a), (double)-sinl(a));
    }
    // HTK filterbank between 20 Hz and min(8000 Hz, Nyquist), equally spaced on the mel scale
    const double fhi = std::min(8000.0, fs / 2.0);
    const double lo = mel_d(20.0), hi = mel_d(fhi);
    double cf[NMEL + 2];
    for (int c = 0; c < NMEL + 2; ++c) cf[c] = lo + (hi - lo) * c / (NMEL + 1);
    std::vector<int> lo_chan(G::NB, -1);
    for (int b = 0; b < G::NB; ++b) {
        const double f = b * df;
        if (f < 20.0 || f > fhi) continue;
        const double m = mel_d(f);
        int c = 0;
        while (c < NMEL && cf[c + 1] <= m) ++c;
        lo_chan[b] = c;
        t.lo_wt[b] = (cf[c + 1] - m) / (cf[c + 1] - cf[c]);
    }
    int b = 0;
    while (b < G::NB && lo_chan[b] < 0) ++b;
    for (int c = 0; c <= NMEL + 1; ++c) {
        while (b < G::NB && lo_chan[b] >= 0 && lo_chan[b] < c) ++b;
        t.seg_start[c] = b;
    }
    // band c (0-based) = rising side (bins whose lower channel is c, weight 1 - lo_wt) + falling side (lower channel c + 1,
    // weight lo_wt); lane 2 c + side walks its side, the weights transposed so that a step of all lanes is one coalesced load
    t.mel_len = 0;
    for (int L = 0; L < 2 * NMEL; ++L) {
        const int seg = (L >> 1) + (L & 1);
        const int b0 = t.seg_start[seg], b1 = t.seg_start[seg + 1];
        t.mel_b0[L] = b0;
        t.mel_len = std::max(t.mel_len, b1 - b0);
        for (int bb = b0; bb < b1 && bb - b0 < G::MELCAP; ++bb)
            t.mel_w[(bb - b0) * 64 + L] = (L & 1) ? t.lo_wt[bb] : 1.0 - t.lo_wt[bb];
    }
    t.mel_len = (t.mel_len + 7) & ~7;                               // the kernel walks the sides 8 bins at a time (zero weights behind a side)
    if (t.mel_len > G::MELCAP) t.mel_len = -1;                      // get_tables refuses
    for (int k = 1; k <= NMFCC; ++k) {
        const double lift = 1.0 + 11.0 * std::sin(M_PI * k / 22.0);
        for (int j = 1; j <= NMEL; ++j)
            t.dct[(j - 1) * 16 + (k - 1)] = std::sqrt(2.0 / NMEL) * std::cos(M_PI * k * (j - 0.5) / NMEL) * lift;
    }
    for (int bb = 0; bb < G::NB; ++bb) {
        const double f = bb * df;
        const double z = 13.0 * std::atan(0.00076 * f) + 3.5 * std::atan((f / 7500.0) * (f / 7500.0));
        const double g = z < 14.0 ? 1.0 : 0.066 * std::exp(0.171 * z);
        t.sharp[bb] = z * g;
    }
    {   // spectral slope: sum f and sum f^2 over bins 0..NC
        double sf = 0.0, sff = 0.0;
        for (int bb = 0; bb < G::NB; ++bb) { sf += bb * df; sff += (bb * df) * (bb * df); }
        t.slope_sf = sf;
        t.slope_den = G::NB * sff - sf * sf;
    }
    // cSpecScale: octave axis from 25 Hz to fs/2 with NB points; natural spline through the equally spaced bins
    const double fmin_l2 = std::log2(25.0), fmax_l2 = std::log2(fs / 2.0);
    const double dl2 = (fmax_l2 - fmin_l2) / (G::NB - 1);
    const double ppo = 1.0 / dl2;
    t.fmin_l2 = fmin_l2;
    t.dl2 = dl2;
    const double atans = ppo * std::log2(65.0 / 50.0) - 1.0;
    for (int i = 0; i < G::NB; ++i) {
        const double pos = std::exp2(fmin_l2 + dl2 * i) / df;
        int k = (int)std::floor(pos);
        if (k > G::NB - 2) k = G::NB - 2;
        t.klo[i] = k;
        t.tb[i] = pos - k;
        t.audw[i] = 0.5 + std::atan(3.0 * (i + 1.0 - atans) / ppo) / M_PI;
    }
    for (int h = 1; h <= NHARM; ++h) {
        t.shs_shift[h - 1] = (int)std::floor(ppo * std::log2((double)h));
        t.shs_w[h - 1] = std::pow(0.85, h - 1);
    }
    // Thomas factors of tridiag(1, 4, 1) on unknowns at bins 1..NC-1 (m_0 = m_NC = 0):
    //   forward  dp_b = g_b (r_b - dp_{b-1}),  backward  x_b = dp_b - g_b x_{b+1},  g_1 = 1/4, g_b = 1 / (4 - g_{b-1})
    double g = 0.0;
    for (int bb = 1; bb < G::NC; ++bb) { g = 1.0 / (4.0 - g); t.sp_g[bb] = g; }
}

static std::mutex g_mu;
static std::map<std::pair<int, int>, void*> g_dev_tables;   // (device, fs) -> device blob

static int log2n_for_frame(int frame) {
    int l = 1;
    while ((1 << l) < frame) ++l;
    return l;
}

static int round_half_up(double x) { return (int)std::floor(x + 0.5); }

int smile_geometry(int fs, int* frame, int* hop, int* log2n) {
    RSAF_CHECK_ARG(fs >= 4000 && fs <= 65536, "sample rate must be in [4000, 65536] Hz");
    const double T = 1.0 / (double)fs;
    *frame = round_half_up(0.025 / T);
    *hop = round_half_up(0.010 / T);
    *log2n = log2n_for_frame(*frame);
    RSAF_CHECK_ARG(*log2n >= 8 && *log2n <= 11, "unsupported frame length for this sample rate");
    return RSAF_OK;
}

template <int LOG2N>
static int get_tables(int fs, int frame, int hop, const Tables<LOG2N>** out) {
    int dev = 0;
    RSAF_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_pair(dev, fs);
    auto it = g_dev_tables.find(key);
    if (it == g_dev_tables.end()) {
        std::vector<Tables<LOG2N>> h(1);
        build_tables<LOG2N>(h[0], fs, frame, hop);
        RSAF_CHECK_ARG(h[0].mel_len >= 0, "mel band wider than the kernel's table");
        RSAF_CHECK_ARG(h[0].shs_shift[NHARM - 1] <= (Geo<LOG2N>::NC * 5) / 8, "harmonic shift longer than the octave spectrum's zero pad");
        void* d = nullptr;
        RSAF_CHECK_HIP(hipMalloc(&d, sizeof(Tables<LOG2N>)));
        RSAF_CHECK_HIP(hipMemcpy(d, h.data(), sizeof(Tables<LOG2N>), hipMemcpyHostToDevice));
        it = g_dev_tables.emplace(key, d).first;
    }
    *out = static_cast<const Tables<LOG2N>*>(it->second);
    return RSAF_OK;
}

// ---- device helpers ----------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ c64 cmul(c64 a, c64 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ c64 cadd(c64 a, c64 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c64 csub(c64 a, c64 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c64 mul_mi(c64 a) { return make_double2(a.y, -a.x); }    // a * (-i)

// a / b by the hardware reciprocal estimate + two Newton steps + one residual correction (error ~1e-16 relative; the
// IEEE division costs about twice as many instructions and none of its callers here decides anything on the last bit)
__device__ __forceinline__ double fdiv(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    const double q = a * y;
    return fma(fma(-b, q, a), y, q);
}

// natural logarithm of a positive normal double: x = m 2^e with m in [sqrt(1/2), sqrt 2), ln m = 2 atanh((m-1)/(m+1)) as an
// odd series in s (|s| <= 0.1716: twelve terms reach 1e-18), ~30 instructions against ~70 for the library routine
__device__ __forceinline__ double flog(double x) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    const double s = fdiv(m - 1.0, m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = fma(p, z, 1.0 / 21.0);
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
    return fma((double)e, LN2, 2.0 * s * p);
}

__device__ __forceinline__ double shfl_f64(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double wave_sum_all(double x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ int wave_min_int(int x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x = min(x, __shfl_xor(x, o, 64));
    return x;
}

// Eight wave-wide sums at once through an LDS transpose: lane writes its 8 partial sums into rows of 65 doubles, lane
// (q = lane >> 3, p = lane & 7) adds 8 entries of row q, three exchanges inside the 8-lane group finish: every lane of
// group q then holds total q.  `red` = 520 doubles of wave-private LDS that nothing else is using.
__device__ __forceinline__ double reduce8(const double (&v)[8], double* red, int lane) {
#pragma unroll
    for (int q = 0; q < 8; ++q) red[q * 65 + lane] = v[q];
    lds_fence();
    const int q = lane >> 3, p = lane & 7;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += red[q * 65 + 8 * p + i];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    lds_fence();
    return s;                                          // total (lane >> 3)
}

// ---- the kernel -------------------------------------------------------------------------------------------------
// time-domain part of one frame: raw samples -> zero crossings, pre-emphasis, Hamming -> the packed-real FFT input in Z;
// returns the lane's partial sums (sum win^2, sum ham win^2, zero crossings)
template <int LOG2N>
__device__ __forceinline__ void frame_to_z(const float* __restrict__ x, const double (&hm)[2 * Geo<LOG2N>::PPL], int frame,
                                           c64* Z, int lane, double& s_w2, double& s_hw2, double& s_zc) {
    using G = Geo<LOG2N>;
    s_w2 = 0.0; s_hw2 = 0.0; s_zc = 0.0;
#pragma unroll
    for (int j = 0; j < G::PPL; ++j) {
        const int n = lane + 64 * j, i0 = 2 * n, i1 = i0 + 1;
        double w0 = 0.0, w1 = 0.0;
        if (i0 < frame) {
            const double xm = i0 > 0 ? (double)x[i0 - 1] : 0.0, x0 = (double)x[i0];
            const double pe = i0 > 0 ? x0 - PREEMPH * xm : x0 * (1.0 - PREEMPH);       // first sample HTK-style
            const double h = hm[2 * j];
            w0 = pe * h;
            s_w2 += w0 * w0;
            s_hw2 += h * w0 * w0;
            if (i0 > 0 && x0 * xm < 0.0) s_zc += 1.0;
            if (i1 < frame) {
                const double x1 = (double)x[i1];
                const double h1 = hm[2 * j + 1];
                w1 = (x1 - PREEMPH * x0) * h1;
                s_w2 += w1 * w1;
                s_hw2 += h1 * w1 * w1;
                if (x1 * x0 < 0.0) s_zc += 1.0;
            }
        }
        Z[n] = make_double2(w0, w1);
    }
    lds_fence();
}

// NC-point complex FFT of Z in place (Stockham autosort: natural order in, natural order out), then the magnitudes of the
// NFFT-point real transform: mag_out[k] = |X[k]|, k = 0 .. NC (wave-private LDS)
// The twiddles of a lane do not depend on the frame: for the 256- and 512-point transforms (<= 1 butterfly per lane and
// stage) they are fetched once per run and stay in registers; the longer transforms read them from the table (L1).
template <int LOG2N>
struct FftTw {
    static constexpr int NC = Geo<LOG2N>::NC, Q = NC / 4;
    static constexpr int BPL = Q >= 64 ? Q / 64 : 1;
    static constexpr int R4 = (LOG2N - 1) / 2;
    static constexpr bool HOIST = LOG2N <= 9;
    c64 w[HOIST ? R4 : 1][3];
    c64 w2;                                                   // radix-2 stage
    c64 wu[HOIST ? Geo<LOG2N>::PPL + 1 : 1];                  // unpack
    __device__ __forceinline__ void load(const Tables<LOG2N>* __restrict__ T, int lane) {
        if constexpr (HOIST) {
            int Ns = 1;
#pragma unroll
            for (int st = 0; st < R4; ++st, Ns *= 4) {
                const int k = lane & (Ns - 1), tstep = (2 * NC) / (4 * Ns);
                w[st][0] = T->tw[k * tstep]; w[st][1] = T->tw[2 * k * tstep]; w[st][2] = T->tw[3 * k * tstep];
            }
            w2 = T->tw[2 * (lane & (NC / 2 - 1))];
#pragma unroll
            for (int j = 0; j <= Geo<LOG2N>::PPL; ++j) wu[j] = T->tw[j < Geo<LOG2N>::PPL ? lane + 64 * j : NC];
        }
    }
};

template <int LOG2N>
__device__ __forceinline__ void fft_mag(c64* Z, const Tables<LOG2N>* __restrict__ T, const FftTw<LOG2N>& W, int lane,
                                        double* __restrict__ mag_out) {
    using G = Geo<LOG2N>;
    constexpr int NC = G::NC, Q = NC / 4;
    constexpr int BPL = Q >= 64 ? Q / 64 : 1;                 // radix-4 butterflies per lane and stage
    constexpr int LOG2NC = LOG2N - 1;
    constexpr int R4 = LOG2NC / 2;                            // radix-4 stages; one radix-2 stage follows when LOG2NC is odd
    constexpr bool HOIST = FftTw<LOG2N>::HOIST;
    int Ns = 1;
#pragma unroll
    for (int st = 0; st < R4; ++st, Ns *= 4) {
        c64 v[BPL][4];
#pragma unroll
        for (int bi = 0; bi < BPL; ++bi) {
            const int b = lane + 64 * bi;
            if (b < Q) {
                const int k = b & (Ns - 1);
                const int tstep = (2 * NC) / (4 * Ns);          // tw index of exp(-2 pi i k / (4 Ns))
                v[bi][0] = Z[b];
                v[bi][1] = Z[b + Q];
                v[bi][2] = Z[b + 2 * Q];
                v[bi][3] = Z[b + 3 * Q];
                if (Ns > 1) {
                    if constexpr (HOIST) {
                        v[bi][1] = cmul(v[bi][1], W.w[st][0]);
                        v[bi][2] = cmul(v[bi][2], W.w[st][1]);
                        v[bi][3] = cmul(v[bi][3], W.w[st][2]);
                    } else {
                        v[bi][1] = cmul(v[bi][1], T->tw[k * tstep]);
                        v[bi][2] = cmul(v[bi][2], T->tw[2 * k * tstep]);
                        v[bi][3] = cmul(v[bi][3], T->tw[3 * k * tstep]);
                    }
                }
            }
        }
        lds_fence();                                           // every read of the stage precedes its writes
#pragma unroll
        for (int bi = 0; bi < BPL; ++bi) {
            const int b = lane + 64 * bi;
            if (b < Q) {
                const int k = b & (Ns - 1);
                const int j0 = ((b - k) << 2) + k;             // (b / Ns) * 4 Ns + k
                const c64 t0 = cadd(v[bi][0], v[bi][2]), t1 = csub(v[bi][0], v[bi][2]);
                const c64 t2 = cadd(v[bi][1], v[bi][3]), t3 = mul_mi(csub(v[bi][1], v[bi][3]));
                Z[j0] = cadd(t0, t2);
                Z[j0 + Ns] = cadd(t1, t3);
                Z[j0 + 2 * Ns] = csub(t0, t2);
                Z[j0 + 3 * Ns] = csub(t1, t3);
            }
        }
        lds_fence();
    }
    if (LOG2NC & 1) {                                          // Ns = NC / 2: one radix-2 stage
        constexpr int H = NC / 2;
        constexpr int B2 = H >= 64 ? H / 64 : 1;
        c64 a[B2], bq[B2];
#pragma unroll
        for (int bi = 0; bi < B2; ++bi) {
            const int b = lane + 64 * bi;
            if (b < H) { a[bi] = Z[b]; bq[bi] = cmul(Z[b + H], HOIST ? W.w2 : T->tw[2 * b]); }     // exp(-2 pi i b / NC)
        }
        lds_fence();
#pragma unroll
        for (int bi = 0; bi < B2; ++bi) {
            const int b = lane + 64 * bi;
            if (b < H) { Z[b] = cadd(a[bi], bq[bi]); Z[b + H] = csub(a[bi], bq[bi]); }
        }
        lds_fence();
    }
    // real-transform bins: X[k] = E + W^k O, E = (Z[k] + conj Z[NC-k]) / 2, O = -i (Z[k] - conj Z[NC-k]) / 2, W = exp(-2 pi i / NFFT)
#pragma unroll
    for (int j = 0; j <= G::PPL; ++j) {
        const int k = j < G::PPL ? lane + 64 * j : NC;
        if (j < G::PPL || lane == 0) {
            const c64 A = Z[k & (NC - 1)], Bc = Z[(NC - k) & (NC - 1)];
            const c64 E = make_double2(0.5 * (A.x + Bc.x), 0.5 * (A.y - Bc.y));
            const c64 D = make_double2(0.5 * (A.x - Bc.x), 0.5 * (A.y + Bc.y));       // (A - conj B) / 2
            const c64 O = cmul(HOIST ? W.wu[HOIST ? j : 0] : T->tw[k], mul_mi(D));
            const double re = E.x + O.x, im = E.y + O.y;
            mag_out[k] = sqrt(re * re + im * im);
        }
    }
    lds_fence();
}

template <int LOG2N>
__global__ __launch_bounds__(64, LOG2N <= 9 ? 3 : 2) void smile_lld_kernel(const float* __restrict__ wav, const int64_t* __restrict__ clip_off,
                                                       const int64_t* __restrict__ frame_off, int64_t total_frames,
                                                       double* __restrict__ lld, double* __restrict__ cand,
                                                       double* __restrict__ octave_dbg,
                                                       const Tables<LOG2N>* __restrict__ T, int stop) {
    using G = Geo<LOG2N>;
    constexpr int NC = G::NC, NB = G::NB, PPL = G::PPL;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane_ = threadIdx.x;
    const int lane = lane_;
    const int clip = blockIdx.y;
    const int64_t fo = frame_off[clip];
    const int n_fr = (int)(frame_off[clip + 1] - fo);
    const int f0 = blockIdx.x * RUNW;
    if (f0 >= n_fr) return;
    const int n_run = min(RUNW, n_fr - f0);
    const float* xclip = wav + clip_off[clip];
    const int frame = T->frame, hop = T->hop;
    const double df = T->df;

    c64* Z = reinterpret_cast<c64*>(smem);
    double* ZD = smem;                                   // the FFT buffer as arrays of doubles once the transform is done
    double* SS = smem + G::OFF_S;                        // second half of the FFT buffer + the zero pad
    double* STASH = smem + G::OFF_STASH;
    double* RED = smem + G::OFF_RED;

    // frame-invariant per-lane constants kept in registers for the run: the FFT twiddles.  (The Hamming window values are
    // re-read per frame and the lane index is redefined per frame through an empty asm: with every lane-only expression
    // hoisted out of the run loop the 512-point instance needed 247 registers - two waves per SIMD - against 166 now.)
    FftTw<LOG2N> W;
    W.load(T, lane);
    double hm[2 * PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) { hm[2 * j] = T->ham[2 * (lane + 64 * j)]; hm[2 * j + 1] = T->ham[2 * (lane + 64 * j) + 1]; }
    if (f0 > 0) {                                        // magnitudes of the frame in front of the run (for the flux): the "previous" array of frame 0
        double a, b, c;
        frame_to_z<LOG2N>(xclip + (int64_t)(f0 - 1) * hop, hm, frame, Z, lane, a, b, c);
        fft_mag<LOG2N>(Z, T, W, lane, smem + G::OFF_P1);
    }
    // the zero pad behind the octave spectrum: outside the FFT buffer and the reduction scratch, written once
    for (int i = G::Z_D - G::OFF_S + lane; i < G::S_D; i += 64) SS[i] = 0.0;

#pragma unroll 1
    for (int tr = 0; tr < n_run; ++tr) {
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        const int fr = f0 + tr;
        const int64_t fg = fo + fr;
        double* MAG = smem + ((tr & 1) ? G::OFF_P1 : G::OFF_P0);      // this frame's magnitudes (the next frame's previous ones)
        double* MAGP = smem + ((tr & 1) ? G::OFF_P0 : G::OFF_P1);     // the previous frame's; from the enhancement on: free
        double s_w2, s_hw2, s_zc;
        double hm[2 * PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) { hm[2 * j] = T->ham[2 * (lane + 64 * j)]; hm[2 * j + 1] = T->ham[2 * (lane + 64 * j) + 1]; }
        frame_to_z<LOG2N>(xclip + (int64_t)fr * hop, hm, frame, Z, lane, s_w2, s_hw2, s_zc);
        fft_mag<LOG2N>(Z, T, W, lane, MAG);
        if (stop == 1) continue;                         // profiling aid (env RSAF_SMILE_STOP): leave the frame after phase k

        // ---- cSpectral partial sums over the lane's bins (power spectrum; Androids.conf:258-280) ----
        double v[8], w[8];
        {
            double tot = 0, pf = 0, sm = 0, b1 = 0, b2 = 0, sh = 0, fx = 0, sl = 0, pl = 0, hrm = 0;
#pragma unroll
            for (int j = 0; j <= PPL; ++j) {
                const int k = j < PPL ? lane + 64 * j : NC;
                if (j == PPL && lane != 0) break;
                const double m = MAG[k], P = m * m, f = (double)k * df;
                tot += P;
                pf += P * f;
                sm += m;
                if (f >= 250.0 && f <= 650.0) b1 += P;
                if (f >= 1000.0 && f <= 4000.0) b2 += P;
                sh += P * T->sharp[k];
                if (fr > 0) { const double d = m - MAGP[k]; fx += d * d; }
                const double lp = flog(fmax(P, 1e-30));
                sl += lp;
                if (P > 0.0) pl += P * lp;                       // P ln P (P < 1e-30 contributes < 1e-28: below every tolerance)
                if (k >= 1 && k < NC) hrm += fmax(m - 0.5 * (MAG[k - 1] + MAG[k + 1]), 0.0);
            }
            v[0] = s_w2; v[1] = s_hw2; v[2] = s_zc; v[3] = tot; v[4] = pf; v[5] = sm; v[6] = b1; v[7] = b2;
            w[0] = sh; w[1] = fx; w[2] = sl; w[3] = pl; w[4] = hrm; w[5] = 0; w[6] = 0; w[7] = 0;
        }
        const double r1 = reduce8(v, RED, lane);
        const double r2 = reduce8(w, RED, lane);
        double* st = STASH + tr * NSUM;
        if ((lane & 7) == 0) { st[lane >> 3] = r1; st[8 + (lane >> 3)] = r2; }
        const double tot = shfl_f64(r1, 24), pfs = shfl_f64(r1, 32);
        const double safe = tot > 0.0 ? tot : 1.0;
        const double cen = pfs / safe;
        {   // central moments about the centroid
            double m2 = 0, m3 = 0, m4 = 0;
#pragma unroll 1
            for (int j = 0; j <= PPL; ++j) {
                const int k = j < PPL ? lane + 64 * j : NC;
                if (j == PPL && lane != 0) break;
                const double mk = MAG[k], P = mk * mk, d = (double)k * df - cen, d2 = d * d;
                m2 += d2 * P; m3 += d2 * d * P; m4 += d2 * d2 * P;
            }
            v[0] = m2; v[1] = m3; v[2] = m4; v[3] = 0; v[4] = 0; v[5] = 0; v[6] = 0; v[7] = 0;
        }
        const double r3 = reduce8(v, RED, lane);
        if ((lane & 7) == 0 && lane < 24) st[16 + (lane >> 3)] = r3;
        if (stop == 2) continue;

        // ---- roll-off points: first bin whose inclusive cumulative power reaches p * total (lane-blocked prefix sums) ----
        {
            double c[PPL];
            double run = 0.0;
#pragma unroll
            for (int j = 0; j < PPL; ++j) { const double m = MAG[PPL * lane + j]; run += m * m; c[j] = run; }
            double incl = run;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const double up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
            const double excl = incl - run;
            const double pr[4] = {0.25, 0.50, 0.75, 0.90};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double th = pr[q] * tot;
                int first = 0x7fffffff;
#pragma unroll
                for (int j = PPL - 1; j >= 0; --j) if (excl + c[j] >= th) first = PPL * lane + j;
                first = wave_min_int(first);
                if (first == 0x7fffffff) first = NC;               // only the last bin is left
                if (lane == 0) st[19 + q] = (double)first * df;
            }
        }

        if (stop == 3) continue;
        // ---- cMelspec + cMfcc (Androids.conf:101-115): lane 2 c + side = one side of triangular band c ----
        {
            double s = 0.0;
            {
                const int b0 = T->mel_b0[lane], nside = T->mel_len;   // lanes >= 52: weights all zero
                for (int i0 = 0; i0 < nside; i0 += 8) {             // 8 coalesced weight loads in flight
                    double wv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) wv[u] = T->mel_w[(i0 + u) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s = fma(wv[u], MAG[min(b0 + i0 + u, NC)], s);
                }
            }
            s += __shfl_xor(s, 1, 64);
            const double lm = flog(fmax(s * HTK_SCALE, MEL_FLOOR));
            if (lane < 2 * NMEL && !(lane & 1)) SS[lane >> 1] = lm;
            lds_fence();
            if (lane < NMFCC) {
                double acc = 0.0;
#pragma unroll
                for (int jj = 0; jj < NMEL; ++jj) acc += T->dct[jj * 16 + lane] * SS[jj];
                lld[(int64_t)(1 + lane) * total_frames + fg] = acc;
            }
            lds_fence();
        }

        if (stop == 4) continue;
        // ---- cSpecScale (Androids.conf:142-160): peak enhancement + (1,2,1) smoothing on the linear spectrum ----
        // (MAG stays as it is: it is the next frame's MAGP)
        double* A1 = ZD;                                   // enhanced spectrum   (FFT buffer, first half)
        double* MM = ZD;                                   // spline moments      (the same half, once the smoothing has consumed A1)
        {
            // local maxima (a[i] > a[i-1] and a[i] >= a[i+1]; the ends count when larger than their one neighbour) as bit
            // masks of 64 bins; "within 2 bins of a maximum" is then shift arithmetic on the scalar unit
            unsigned long long mx[PPL + 1];
#pragma unroll
            for (int j = 0; j <= PPL; ++j) {
                const int k = lane + 64 * j;
                bool is = false;
                if (k <= NC) {
                    const double a = MAG[k];
                    if (k == 0) is = a > MAG[1];
                    else if (k == NC) is = a > MAG[NC - 1];
                    else is = a > MAG[k - 1] && a >= MAG[k + 1];
                }
                mx[j] = __ballot(is);
            }
            int count = 0, first = -1, last = -1;
#pragma unroll
            for (int j = 0; j <= PPL; ++j) {
                count += __popcll(mx[j]);
                if (mx[j]) {
                    if (first < 0) first = 64 * j + __ffsll((long long)mx[j]) - 1;
                    last = 64 * j + 63 - __clzll((long long)mx[j]);
                }
            }
#pragma unroll
            for (int j = 0; j <= PPL; ++j) {
                const int k = lane + 64 * j;
                if (k > NC) break;
                unsigned long long near = mx[j] | (mx[j] << 1) | (mx[j] << 2) | (mx[j] >> 1) | (mx[j] >> 2);
                if (j > 0) near |= (mx[j - 1] >> 63) | (mx[j - 1] >> 62);             // maxima at bins 64 j - 1, 64 j - 2
                if (j < PPL) near |= (mx[j + 1] << 63) | (mx[j + 1] << 62);           // ... at bins 64 (j + 1), 64 (j + 1) + 1
                const bool nr = (near >> lane) & 1ull;
                double a = MAG[k];
                if (count == 1) { if (!nr) a = 0.0; }
                else if (count > 1) { if (!nr && k > first && k < last) a = 0.0; }
                A1[k] = a;
            }
            lds_fence();
            // smoothing -> A2 (the previous frame's magnitude array: the flux was its last reader)
#pragma unroll 1
            for (int j = 0; j <= PPL; ++j) {
                const int k = lane + 64 * j;
                if (k < NC) MAGP[k] = ((k > 0 ? A1[k - 1] : 0.0) + 2.0 * A1[k] + A1[k + 1]) / 4.0;
                else if (k == NC) MAGP[k] = A1[NC];
            }
            lds_fence();
        }
        if (stop == 5) continue;
        double* A2 = MAGP;
        {   // natural cubic spline through the bins: tridiag(1, 4, 1) m = second differences, m_0 = m_NC = 0.
            // Lane-blocked Thomas algorithm: the forward and the backward recurrence are affine maps x -> A x + B per bin;
            // a lane composes its PPL bins, a wave scan composes the lanes, the carry then re-runs the lane's bins.
            double g[PPL], r[PPL], dp[PPL];
            double Am = 1.0, Bm = 0.0;
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const int b = PPL * lane + j;
                g[j] = T->sp_g[b];                                  // 0 at bin 0
                r[j] = b >= 1 ? (A2[b - 1] - 2.0 * A2[b]) + A2[b + 1] : 0.0;
                // dp_b = -g dp_{b-1} + g r
                Bm = fma(-g[j], Bm, g[j] * r[j]);
                Am = -g[j] * Am;
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double Au = __shfl_up(Am, o, 64), Bu = __shfl_up(Bm, o, 64);
                if (lane >= o) { Bm = fma(Am, Bu, Bm); Am = Am * Au; }
            }
            double carry = __shfl_up(Bm, 1, 64);
            if (lane == 0) carry = 0.0;
#pragma unroll
            for (int j = 0; j < PPL; ++j) { carry = g[j] * (r[j] - carry); dp[j] = carry; }
            // backward: x_b = dp_b - g_b x_{b+1}, x_NC = 0
            Am = 1.0; Bm = 0.0;
#pragma unroll
            for (int j = PPL - 1; j >= 0; --j) { Bm = fma(-g[j], Bm, dp[j]); Am = -g[j] * Am; }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double Ad = __shfl_down(Am, o, 64), Bd = __shfl_down(Bm, o, 64);
                if (lane + o < 64) { Bm = fma(Am, Bd, Bm); Am = Am * Ad; }
            }
            carry = __shfl_down(Bm, 1, 64);
            if (lane == 63) carry = 0.0;
#pragma unroll
            for (int j = PPL - 1; j >= 0; --j) { carry = fma(-g[j], carry, dp[j]); MM[PPL * lane + j] = carry; }
            if (lane == 0) MM[NC] = 0.0;
            lds_fence();
        }
        if (stop == 6) continue;
        // spline at the octave-scale targets, negatives reset, auditory weighting -> S
#pragma unroll 1
        for (int j = 0; j <= PPL; ++j) {
            const int i = lane + 64 * j;
            if (i > NC) break;
            const int kl = T->klo[i];
            const double b = T->tb[i], aa = 1.0 - b;
            const double y = aa * A2[kl] + b * A2[kl + 1] + (aa * aa * aa - aa) * MM[kl] + (b * b * b - b) * MM[kl + 1];
            const double s = fmax(y, 0.0) * T->audw[i];
            SS[i] = s;
            if (octave_dbg) octave_dbg[fg * NB + i] = s;
        }
        lds_fence();

        if (stop == 7) continue;
        // ---- cPitchShs (Androids.conf:162-186): sub-harmonic summation, peaks, the 6 best candidates ----
        double* HH = MAGP;                                  // A2 is dead (spline and evaluation are done): the summation spectrum
        double hsum = 0.0;
#pragma unroll 1
        for (int j = 0; j <= PPL; ++j) {
            const int i = lane + 64 * j;
            if (i <= NC) {
                double acc = 0.0;
#pragma unroll
                for (int h = 0; h < NHARM; ++h) acc += T->shs_w[h] * SS[i + T->shs_shift[h]];   // wave-uniform table reads; zeros behind bin NC
                HH[i] = acc;
                hsum += acc;
            }
        }
        const double hmean = wave_sum_all(hsum) / (double)NB;
        lds_fence();
        if (stop == 8) continue;
        // peaks (y2 > y1 and y2 >= y3) with parabolic refinement, inside 52..620 Hz, positive score: compacted into a list
        double* L_sc = SS;                                  // S is dead: list of (score, f0); at most NC / 2 peaks, the pad stays zero
        double* L_f = SS + NC / 2;
        int npk = 0;
#pragma unroll 1
        for (int j = 0; j <= PPL; ++j) {
            const int i = lane + 64 * j;
            bool ok = false;
            double sc = 0.0, fq = 0.0;
            if (i >= 1 && i <= NC - 1) {
                const double y1 = HH[i - 1], y2 = HH[i], y3 = HH[i + 1];
                if (y2 > y1 && y2 >= y3) {
                    const double den = (y1 - 2.0 * y2) + y3;
                    const double dx = 0.5 * (y1 - y3) / den;
                    sc = y2 - 0.125 * (y1 - y3) * (y1 - y3) / den;
                    fq = exp2(T->fmin_l2 + ((double)i + dx) * T->dl2);
                    ok = fq >= SHS_MINPITCH && fq <= SHS_MAXPITCH && sc > 0.0;
                }
            }
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const int p = npk + __popcll(m & ((1ull << lane) - 1ull));
                L_sc[p] = sc;
                L_f[p] = fq;
            }
            npk += __popcll(m);
        }
        lds_fence();
        if (stop == 9) continue;
        // rank = number of peaks with a higher score (ties: the lower index, i.e. the earlier list entry, first)
        double* cd = cand + fg * (NCAND * 2);
        if (lane < NCAND && lane >= npk) { cd[2 * lane] = 0.0; cd[2 * lane + 1] = 0.0; }
        for (int p0 = 0; p0 < npk; p0 += 64) {
            const int p = p0 + lane;
            const double mine = p < npk ? L_sc[p] : 0.0;
            int rank = 0;
            for (int q = 0; q < npk; ++q) {
                const double o = L_sc[q];
                rank += (o > mine || (o == mine && q < p)) ? 1 : 0;
            }
            if (p < npk && rank < NCAND) {
                cd[2 * rank] = L_f[p];
                cd[2 * rank + 1] = fmax(0.0, 1.0 - hmean / mine);
            }
        }
        lds_fence();
    }

    // ---- end of the run: lane t finishes frame f0 + t from its sums ----
    if (lane < n_run) {
        const double* st = STASH + lane * NSUM;
        const int64_t fg = fo + f0 + lane;
        const double N = (double)frame, NBd = (double)NB;
        auto out = [&](int row, double val) { lld[(int64_t)row * total_frames + fg] = val; };
        const double tot = st[3], safe = tot > 0.0 ? tot : 1.0;
        out(0, sqrt(st[0] / N));                                             // pcm_RMSenergy (cEnergy on winframe)
        out(13, st[2] / N);                                                  // pcm_zcr (cMZcr on the raw frame)
        const double inten = (st[1] / T->ham_sum) / I0;
        out(16, inten);                                                      // pcm_intensity
        out(17, pow(inten, 0.3));                                            // pcm_loudness
        out(22, st[6]);
        out(23, st[7]);
        out(24, st[19]); out(25, st[20]); out(26, st[21]); out(27, st[22]);  // roll-off 25 / 50 / 75 / 90 %
        out(28, sqrt(st[9] / NBd));                                          // flux (0 for the first frame of a clip)
        out(29, st[4] / safe);                                               // centroid
        out(30, tot > 0.0 ? log2(tot) - (st[11] / LN2) / tot : 0.0);         // entropy = -sum p log2 p, p = P / total
        const double var = st[16] / safe, vs = var > 0.0 ? var : 1.0;
        out(31, var);
        out(32, (st[17] / safe) / (vs * sqrt(vs)));
        out(33, (st[18] / safe) / (vs * vs));
        out(34, (NBd * st[4] - T->slope_sf * tot) / T->slope_den);           // slope of the power spectrum over frequency
        out(35, st[8] / safe);                                               // psychoacoustic sharpness
        out(36, st[12] / (st[5] > 0.0 ? st[5] : 1.0));                       // harmonicity proxy
        // flatness = geometric / arithmetic mean of the power spectrum; a frame of zeros has the limit value 1
        out(37, tot > 0.0 ? exp(st[10] / NBd) / fmax(tot / NBd, 1e-30) : 1.0);
    }
}

template <int LOG2N>
static int launch(const float* wav, const int64_t* clip_off, const int64_t* frame_off, int n_clips,
                  int64_t max_clip_frames, int64_t total_frames, int fs, int frame, int hop, double* lld, double* cand,
                  double* octave_dbg, hipStream_t s) {
    using G = Geo<LOG2N>;
    const Tables<LOG2N>* tab = nullptr;
    int rc = get_tables<LOG2N>(fs, frame, hop, &tab);
    if (rc != RSAF_OK) return rc;
    const int64_t runs = (max_clip_frames + RUNW - 1) / RUNW;
    RSAF_CHECK_ARG(runs <= 0x7fffffffLL, "clip too long");
    constexpr size_t lds = (size_t)G::WAVE_D * sizeof(double);
    static_assert(lds <= 64 * 1024, "one wave's buffers must fit the default dynamic LDS limit");
    static_assert(LOG2N < 11 || lds <= 40 * 1024, "the 2 048-point instance must leave room for four waves per CU");
    // algorithmic bytes: every sample read once (4 B) + the LLD rows written (38 * 8 B per frame)
    ProfScope prof("smile_lld", s, 0.0, 0.0);
    dim3 grid((unsigned)runs, (unsigned)n_clips);
    static const int stop = [] { const char* e = getenv("RSAF_SMILE_STOP"); return e ? atoi(e) : 0; }();
    hipLaunchKernelGGL(smile_lld_kernel<LOG2N>, grid, dim3(64), lds, s, wav, clip_off, frame_off, total_frames, lld, cand,
                       octave_dbg, tab, stop);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

}  // namespace smile
}  // namespace rsaf

using namespace rsaf;
using namespace rsaf::smile;

extern "C" {

int rsaf_smile_geometry(int sample_rate, int* frame_host, int* hop_host, int* nfft_host) {
    RSAF_CHECK_ARG(frame_host && hop_host && nfft_host, "NULL output");
    int l2 = 0;
    int rc = smile_geometry(sample_rate, frame_host, hop_host, &l2);
    if (rc != RSAF_OK) return rc;
    *nfft_host = 1 << l2;
    return RSAF_OK;
}

int64_t rsaf_smile_n_frames(int64_t n_samples, int sample_rate) {
    int frame = 0, hop = 0, l2 = 0;
    if (smile_geometry(sample_rate, &frame, &hop, &l2) != RSAF_OK) return -1;
    return n_samples < frame ? 0 : (n_samples - frame) / hop + 1;
}

int rsaf_init_device(int device) {
    RSAF_CHECK_HIP(hipSetDevice(device));
    const Tables<9>* t = nullptr;
    return get_tables<9>(16000, 400, 160, &t);
}

int rsaf_smile_lld_batch(const float* wav, const int64_t* clip_off, const int64_t* frame_off,
                         int n_clips, int64_t max_clip_frames, int64_t total_frames, int sample_rate, double* lld,
                         double* cand, double* octave_spectrum, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && n_clips <= 65535, "n_clips must be in [0, 65535] per call");
    RSAF_CHECK_ARG(total_frames >= 0 && max_clip_frames >= 0, "negative frame count");
    int frame = 0, hop = 0, l2 = 0;
    int rc = smile_geometry(sample_rate, &frame, &hop, &l2);
    if (rc != RSAF_OK) return rc;
    if (n_clips == 0 || total_frames == 0 || max_clip_frames == 0) return RSAF_OK;
    RSAF_CHECK_ARG(wav && clip_off && frame_off && lld && cand, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    switch (l2) {
        case 8: return launch<8>(wav, clip_off, frame_off, n_clips, max_clip_frames, total_frames, sample_rate, frame, hop, lld, cand, octave_spectrum, s);
        case 9: return launch<9>(wav, clip_off, frame_off, n_clips, max_clip_frames, total_frames, sample_rate, frame, hop, lld, cand, octave_spectrum, s);
        case 10: return launch<10>(wav, clip_off, frame_off, n_clips, max_clip_frames, total_frames, sample_rate, frame, hop, lld, cand, octave_spectrum, s);
        default: return launch<11>(wav, clip_off, frame_off, n_clips, max_clip_frames, total_frames, sample_rate, frame, hop, lld, cand, octave_spectrum, s);
    }
}

}  // extern "C"
