// openSMILE-style low-level descriptors for gfx950 (wave64), at the file's own sample rate.
//
// One fused kernel for Androids.conf:73-186 and :258-280 of the reference
// (cFramer -> cVectorPreemphasis -> cWindower -> cTransformFFT -> cFFTmagphase ->
//  {cMelspec -> cMfcc, cEnergy, cMZcr, cIntensity, cSpectral, cSpecScale -> cPitchShs}); the reference runs that
// chain by spawning SMILExtract once per file (src/opensmile_extractor.py:62-75).  The sequential tail of the
// pitch chain (cPitchSmootherViterbi, cValbasedSelector, cPitchJitter) is csrc/smile_pitch.hip.
//
// Mapping.  A wave owns PAIRS of consecutive frames and keeps the two frames in the two halves of packed
// float2 registers: every elementwise step (pre-emphasis, window, FFT butterflies, spectral descriptors,
// spline, sub-harmonic summation) is then v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on both frames at once,
// and every LDS access moves 8 or 16 bytes per lane.  The kernel is VALU-issue bound (profiles/r02), so this is
// the lever: half the vector instructions per frame.  A workgroup of 8 waves (4 for the 2048-point FFT) shares
// one LDS copy of the constant tables (twiddles, Hamming, mel weights, spline / octave-scale tables); each wave
// stages its own samples (the 2.5x frame overlap is served by L2, HBM sees every sample once), and has a
// private FFT buffer that is reused for the octave-spectrum arrays.  FFT: packed-real N-point transform as an
// N/2-point complex Stockham radix-4 (+ one radix-2 stage when log2(N/2) is odd), in place in LDS (all reads of a
// stage precede its writes in program order; a wave is lock-step, so no second buffer is needed).
// Templated on the FFT length (256 / 512 / 1024 / 2048 <-> 8 / 16 / 22.05-32 / 44.1-48 kHz; frame and hop
// lengths are run-time).
//
// Spectral flux needs the previous frame's magnitudes: inside a wave's span they are in registers; for its first
// frame the wave transforms the frame in front of its span once more (window + FFT only: +6 % work), which keeps
// the waves independent of each other.
//
// Semantics are those of oracle/smile_oracle.py (the CPU restatement), parity unpinned.
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "rsaf_common.h"

namespace rsaf {
namespace smile {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NLLD = RSAF_SMILE_NLLD;
constexpr int NMEL = 26;
constexpr int NMFCC = 12;
constexpr int NCAND = RSAF_SMILE_NCAND;
constexpr int NHARM = 15;
constexpr int NLOCAL = 32;               // LLD rows this kernel produces (38 minus the six pitch-chain rows)
constexpr float PREEMPH = 0.97f;
constexpr float HTK_SCALE = 32767.0f;
constexpr float MEL_FLOOR = 1.0f;

template <int LOG2N>
struct Geo {
    static constexpr int NFFT = 1 << LOG2N;
    static constexpr int NC = NFFT / 2;              // complex points of the packed-real transform
    static constexpr int NB = NC + 1;                // magnitude bins
    static constexpr int PPL = NC / 64;              // bins per lane
    static constexpr int NBP = NC + 8;               // padded length of the per-bin tables
    // lanes of carry kept in the lane-blocked tridiagonal solve: a lane block damps a carry by 0.268^PPL, so
    // 4 lanes reach 0.268^12 ~ 1e-7 of a term that is itself < 0.27 of the local one (PPL = 2: 6 lanes)
    static constexpr int CARRY = PPL >= 4 ? 4 : 6;
    static constexpr int HALF = NC + 8 + (NC + 8) / 32 + 1;   // float2 entries of one half of the wave's FFT buffer (swizzled)
    static constexpr int XF = 4 * HALF;              // floats of the wave's FFT buffer
    static constexpr int MF = 2 * HALF;              // floats of the wave's magnitude slot
    static constexpr int WAVES = LOG2N >= 11 ? 4 : 8;
    static constexpr int PPW = LOG2N <= 9 ? 4 : 2;   // frame pairs per wave
    static constexpr int RUN = WAVES * PPW * 2;      // frames per workgroup
};

// ---- constant tables (host-computed in double, one blob per (device, sample rate)) -------------------
template <int LOG2N>
struct __attribute__((aligned(16))) Tables {
    using G = Geo<LOG2N>;
    float2 twr[G::NC + G::NC / 2];    // exp(-2 pi i k / NFFT), k < 3 NC / 2: the packed-real unpack uses k < NC, the
                                      // complex stages exp(-2 pi i m / NC) = twr[2 m], m < 3 NC / 4
    float ham[G::NFFT];               // Hamming window, zero beyond the frame
    float lo_wt[G::NBP];              // HTK lower-channel weight per bin (0 where unused)
    float sharp[G::NBP];              // bark(f) * g(bark) per bin
    float tb[G::NBP];                 // octave-scale target i: fractional position inside source interval klo[i]
    float audw[G::NBP];               // auditory weighting of target i
    int klo[G::NBP];
    float sp_g[G::NBP];               // tridiagonal (1, 4, 1) elimination factors, 0 at bin 0
    float sp_cf[G::CARRY - 1][64];    // forward / backward carry coefficients of the lane-blocked solve
    float sp_cb[G::CARRY - 1][64];
    float dct[NMFCC * NMEL];          // DCT-II rows 1..12 with the lifter folded in
    int seg_start[32];                // bins with lower channel c are [seg_start[c], seg_start[c+1]), c = 0..26
    int mel_lane[64];                 // lane's chunk of a band side: start bin | length << 12 | rising << 20 (0 = idle)
    int mel_band[32];                 // band c-1: first lane | number of lanes << 8
    int shs_shift[16];
    float shs_w[16];
    float ham_sum, df, fmin_l2, dl2;
    float band1_lo, band1_hi, band2_lo, band2_hi;
    float slope_sf, slope_den, pad0, pad1;
    int frame, hop, fs, max_seg;
    int mel_iters, mel_max_n, pad2, pad3;
};

static double mel_d(double f) { return 2595.0 * std::log10(1.0 + f / 700.0); }

template <int LOG2N>
static void build_tables(Tables<LOG2N>& t, int fs, int frame, int hop) {
    using G = Geo<LOG2N>;
    std::memset(&t, 0, sizeof(t));
    t.fs = fs; t.frame = frame; t.hop = hop;
    const double df = (double)fs / G::NFFT;
    t.df = (float)df;
    double hs = 0;
    for (int i = 0; i < frame; ++i) {
        const double w = 0.54 - 0.46 * std::cos(2.0 * M_PI * i / (frame - 1));
        t.ham[i] = (float)w;
        hs += w;
    }
    t.ham_sum = (float)hs;
    for (int m = 0; m < G::NC + G::NC / 2; ++m)
        t.twr[m] = make_float2((float)std::cos(2.0 * M_PI * m / G::NFFT), (float)-std::sin(2.0 * M_PI * m / G::NFFT));
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
        t.lo_wt[b] = (float)((cf[c + 1] - m) / (cf[c + 1] - cf[c]));
    }
    int b = 0;
    while (b < G::NB && lo_chan[b] < 0) ++b;
    int max_seg = 0;
    for (int c = 0; c <= NMEL + 1; ++c) {
        while (b < G::NB && lo_chan[b] >= 0 && lo_chan[b] < c) ++b;
        t.seg_start[c] = b;
        if (c > 0) max_seg = std::max(max_seg, t.seg_start[c] - t.seg_start[c - 1]);
    }
    t.max_seg = max_seg;
    // lane-balanced schedule of the 52 band sides (falling side of band c = bins with lower channel c weighted lo_wt,
    // rising side = bins with lower channel c-1 weighted 1 - lo_wt): chunks of at most CH bins, one chunk per lane
    for (int CH = 1; CH <= G::NB; ++CH) {
        int need = 0;
        for (int c = 1; c <= NMEL; ++c)
            for (int side = 0; side < 2; ++side) {
                const int seg = side ? c - 1 : c;
                const int L = t.seg_start[seg + 1] - t.seg_start[seg];
                need += (L + CH - 1) / CH;
            }
        if (need > 64) continue;
        int lane = 0, max_n = 0;
        for (int c = 1; c <= NMEL; ++c) {
            const int first = lane;
            for (int side = 0; side < 2; ++side) {
                const int seg = side ? c - 1 : c;
                for (int b0 = t.seg_start[seg]; b0 < t.seg_start[seg + 1]; b0 += CH) {
                    const int L = std::min(CH, t.seg_start[seg + 1] - b0);
                    t.mel_lane[lane++] = b0 | (L << 12) | (side << 20);
                }
            }
            t.mel_band[c - 1] = first | ((lane - first) << 8);
            max_n = std::max(max_n, lane - first);
        }
        t.mel_iters = CH;
        t.mel_max_n = max_n;
        break;
    }
    for (int k = 1; k <= NMFCC; ++k) {
        const double lift = 1.0 + 11.0 * std::sin(M_PI * k / 22.0);
        for (int j = 1; j <= NMEL; ++j)
            t.dct[(k - 1) * NMEL + (j - 1)] = (float)(std::sqrt(2.0 / NMEL) * std::cos(M_PI * k * (j - 0.5) / NMEL) * lift);
    }
    for (int bb = 0; bb < G::NB; ++bb) {
        const double f = bb * df;
        const double z = 13.0 * std::atan(0.00076 * f) + 3.5 * std::atan((f / 7500.0) * (f / 7500.0));
        const double g = z < 14.0 ? 1.0 : 0.066 * std::exp(0.171 * z);
        t.sharp[bb] = (float)(z * g);
    }
    t.band1_lo = 250.f; t.band1_hi = 650.f; t.band2_lo = 1000.f; t.band2_hi = 4000.f;
    {   // spectral slope: sum f and sum f^2 over bins 0..NC
        const double n = G::NC;
        const double sf = df * n * (n + 1) / 2.0, sff = df * df * n * (n + 1) * (2 * n + 1) / 6.0;
        t.slope_sf = (float)sf;
        t.slope_den = (float)(G::NB * sff - sf * sf);
    }
    // cSpecScale: octave axis from 25 Hz to fs/2 with NB points; natural spline through the equally spaced bins
    const double fmin_l2 = std::log2(25.0), fmax_l2 = std::log2(fs / 2.0);
    const double dl2 = (fmax_l2 - fmin_l2) / (G::NB - 1);
    const double ppo = 1.0 / dl2;
    t.fmin_l2 = (float)fmin_l2;
    t.dl2 = (float)dl2;
    const double atans = ppo * std::log2(65.0 / 50.0) - 1.0;
    for (int i = 0; i < G::NB; ++i) {
        const double pos = std::exp2(fmin_l2 + dl2 * i) / df;
        int k = (int)std::floor(pos);
        if (k > G::NB - 2) k = G::NB - 2;
        t.klo[i] = k;
        t.tb[i] = (float)(pos - k);
        t.audw[i] = (float)(0.5 + std::atan(3.0 * (i + 1.0 - atans) / ppo) / M_PI);
    }
    for (int h = 1; h <= NHARM; ++h) {
        t.shs_shift[h - 1] = (int)std::floor(ppo * std::log2((double)h));
        t.shs_w[h - 1] = (float)std::pow(0.85, h - 1);
    }
    // Thomas factors of tridiag(1, 4, 1) on unknowns at bins 1..NC-1 (m_0 = m_NC = 0):
    //   forward  dp_b = g_b (r_b - dp_{b-1}),  backward  x_b = dp_b - g_b x_{b+1},  g_1 = 1/4, g_b = 1 / (4 - g_{b-1})
    std::vector<double> g(G::NC + 1, 0.0), P(64, 1.0);
    for (int bb = 1; bb < G::NC; ++bb) g[bb] = 1.0 / (4.0 - g[bb - 1]);
    for (int bb = 0; bb < G::NC; ++bb) t.sp_g[bb] = (float)g[bb];
    for (int L = 0; L < 64; ++L)
        for (int i = 0; i < G::PPL; ++i) P[L] *= -g[G::PPL * L + i];
    for (int L = 0; L < 64; ++L) {
        double cf_ = 1.0, cb_ = 1.0;
        for (int d = 2; d <= G::CARRY; ++d) {
            cf_ *= (L - (d - 1) >= 0) ? P[L - (d - 1)] : 0.0;
            cb_ *= (L + (d - 1) < 64) ? P[L + (d - 1)] : 0.0;
            t.sp_cf[d - 2][L] = (float)cf_;
            t.sp_cb[d - 2][L] = (float)cb_;
        }
    }
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

__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
// LDS index of entry b of a float2 array read PPL-consecutive-per-lane (32 B lane stride): one pad entry per 32 spreads
// the 32 lanes of a b64 access group over all 64 banks (plain indexing is a 4-way conflict)
__device__ __forceinline__ int sw(int b) { return b + (b >> 5); }
__device__ __forceinline__ v2f vmax2(v2f a, v2f b) { return (v2f){fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ v2f wave_sum2(v2f a) { return (v2f){wave_sum(a.x), wave_sum(a.y)}; }
// 1-ulp hardware forms (the IEEE expansions of /, sqrtf, logf cost ~10 instructions each; the parity bar is 1e-4)
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float flog(float x) { return flog2(x) * 0.6931471805599453f; }
__device__ __forceinline__ v2f frcp2(v2f a) { return (v2f){frcp(a.x), frcp(a.y)}; }
__device__ __forceinline__ v2f fsqrt2(v2f a) { return (v2f){fsqrt(a.x), fsqrt(a.y)}; }
__device__ __forceinline__ v2f flog2_2(v2f a) { return (v2f){flog2(a.x), flog2(a.y)}; }
__device__ __forceinline__ v2f fexp2_2(v2f a) { return (v2f){fexp2(a.x), fexp2(a.y)}; }

// Eight wave reductions at once (gfx950 lane swaps): v_permlane32_swap / v_permlane16_swap fold two registers into
// one whose halves / rows hold different quantities, then three DPP steps finish inside 8-lane groups: 18 vector
// instructions + 8 v_readlane for eight totals instead of 8 x 7.  v[] comes back wave-uniform.
template <bool IS_MIN>
__device__ __forceinline__ void wave_reduce8(float (&v)[8]) {
    auto op = [](float a, float b) { return IS_MIN ? fminf(a, b) : a + b; };
    float a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
        a[i] = op(__uint_as_float(r[0]), __uint_as_float(r[1]));           // lanes < 32: v[i], lanes >= 32: v[i+4]
    }
    float b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[2 * i]), __float_as_uint(a[2 * i + 1]), false, false);
        b[i] = op(__uint_as_float(r[0]), __uint_as_float(r[1]));           // rows: v[2i], v[2i+1], v[2i+4], v[2i+5]
    }
    const bool hi8 = (threadIdx.x & 8) != 0;
    const float keep = hi8 ? b[1] : b[0], give = hi8 ? b[0] : b[1];
    float z = op(keep, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give), 0x128, 0xF, 0xF, false)));   // row_ror:8
    z = op(z, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(z), 0xB1, 0xF, 0xF, false)));
    z = op(z, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(z), 0x4E, 0xF, 0xF, false)));
    z = op(z, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(z), 0x141, 0xF, 0xF, false)));             // row_half_mirror
    // row r, half h holds: (r0: v0 | v2) (r1: v1 | v3) (r2: v4 | v6) (r3: v5 | v7)
    v[0] = readlane_f32(z, 0);  v[2] = readlane_f32(z, 8);
    v[1] = readlane_f32(z, 16); v[3] = readlane_f32(z, 24);
    v[4] = readlane_f32(z, 32); v[6] = readlane_f32(z, 40);
    v[5] = readlane_f32(z, 48); v[7] = readlane_f32(z, 56);
}
__device__ __forceinline__ void wave_sum4x2(v2f& a, v2f& b, v2f& c, v2f& d) {
    float v[8] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
    wave_reduce8<false>(v);
    a = (v2f){v[0], v[1]}; b = (v2f){v[2], v[3]}; c = (v2f){v[4], v[5]}; d = (v2f){v[6], v[7]};
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {      // zero fill of the shifts is neutral
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ float wave_shr1(float v) { return dpp_f32<0x138>(v); }   // lane l <- lane l-1, 0 into lane 0
__device__ __forceinline__ float wave_shl1(float v) { return dpp_f32<0x130>(v); }   // lane l <- lane l+1, 0 into lane 63

// complex multiply of a packed pair (re, im) by the twiddle (c, s)
__device__ __forceinline__ void cmul2(v2f& re, v2f& im, float2 w) {
    const v2f r = re * w.x - im * w.y;
    const v2f i = re * w.y + im * w.x;
    re = r; im = i;
}

struct C2 { v2f re, im; };

__device__ __forceinline__ void radix4(C2& v0, C2& v1, C2& v2, C2& v3) {
    const C2 a0{v0.re + v2.re, v0.im + v2.im};
    const C2 a1{v0.re - v2.re, v0.im - v2.im};
    const C2 a2{v1.re + v3.re, v1.im + v3.im};
    const C2 t{v1.re - v3.re, v1.im - v3.im};
    const C2 a3{t.im, -t.re};                       // -i * t
    v0 = C2{a0.re + a2.re, a0.im + a2.im};
    v1 = C2{a1.re + a3.re, a1.im + a3.im};
    v2 = C2{a0.re - a2.re, a0.im - a2.im};
    v3 = C2{a1.re - a3.re, a1.im - a3.im};
}

__device__ __forceinline__ C2 ldc(const v4f* X, int i) { const v4f v = X[i]; return C2{(v2f){v.x, v.y}, (v2f){v.z, v.w}}; }
__device__ __forceinline__ void stc(v4f* X, int i, C2 c) { X[i] = (v4f){c.re.x, c.re.y, c.im.x, c.im.y}; }

// one Stockham radix-4 stage in place: all butterflies of the wave are read, then written
template <int NC, int NS>
__device__ __forceinline__ void fft_stage4(v4f* X, const float2* __restrict__ tw, int lane) {
    constexpr int NBF = NC / 4;
    constexpr int U = (NBF + 63) / 64;
    C2 v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = lane + 64 * u;
        if (NBF >= 64 || j < NBF) {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[u][t] = ldc(X, j + t * NBF);
            if (NS > 1) {
                const int m = (j & (NS - 1)) * (NC / (NS * 4));
                cmul2(v[u][1].re, v[u][1].im, tw[2 * m]);             // tw = exp(-2 pi i k / (2 NC)): every other entry
                cmul2(v[u][2].re, v[u][2].im, tw[4 * m]);
                cmul2(v[u][3].re, v[u][3].im, tw[6 * m]);
            }
            radix4(v[u][0], v[u][1], v[u][2], v[u][3]);
        }
    }
    lds_fence();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = lane + 64 * u;
        if (NBF >= 64 || j < NBF) {
            const int k = j & (NS - 1);
            const int j0 = ((j - k) << 2) + k;
#pragma unroll
            for (int t = 0; t < 4; ++t) stc(X, j0 + t * NS, v[u][t]);
        }
    }
    lds_fence();
}

template <int NC, int NS>
__device__ __forceinline__ void fft_stage2(v4f* X, const float2* __restrict__ tw, int lane) {
    constexpr int NBF = NC / 2;
    constexpr int U = (NBF + 63) / 64;
    C2 v[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = lane + 64 * u;
        v[u][0] = ldc(X, j);
        v[u][1] = ldc(X, j + NBF);
        const int m = (j & (NS - 1)) * (NC / (NS * 2));
        cmul2(v[u][1].re, v[u][1].im, tw[2 * m]);
        const C2 a{v[u][0].re + v[u][1].re, v[u][0].im + v[u][1].im};
        const C2 b{v[u][0].re - v[u][1].re, v[u][0].im - v[u][1].im};
        v[u][0] = a; v[u][1] = b;
    }
    lds_fence();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = lane + 64 * u;
        const int k = j & (NS - 1);
        const int j0 = ((j - k) << 1) + k;
        stc(X, j0, v[u][0]);
        stc(X, j0 + NS, v[u][1]);
    }
    lds_fence();
}

// stages after the first one (which takes its input from registers)
template <int NC, int NS>
__device__ __forceinline__ void fft_rest(v4f* X, const float2* __restrict__ tw, int lane) {
    if constexpr (NS * 4 <= NC) {
        fft_stage4<NC, NS>(X, tw, lane);
        fft_rest<NC, NS * 4>(X, tw, lane);
    } else if constexpr (NS * 2 <= NC) {
        fft_stage2<NC, NS>(X, tw, lane);
    }
}

template <int LOG2N>
struct Smem {
    using G = Geo<LOG2N>;
    Tables<LOG2N> tab;
    float x[G::WAVES][G::XF];             // per-wave FFT buffer, later [a | m], [S | H] (two halves of HALF float2)
    float mag[G::WAVES][G::MF];           // per-wave magnitudes of the current pair (float2 per bin)
    float melbuf[G::WAVES][180];          // per-wave: 64 float2 chunk sums, then 26 float2 log mel energies
    float out[NLOCAL][G::RUN];            // LLD rows of this run
};

// local row of an LLD index (the six pitch-chain rows 14, 15, 18..21 are written by smile_pitch.hip)
__host__ __device__ constexpr int local_row(int lld) { return lld < 14 ? lld : (lld < 18 ? lld - 2 : lld - 6); }
__host__ __device__ constexpr int lld_of_local(int r) { return r < 14 ? r : (r < 16 ? r + 2 : r + 6); }

template <int LOG2N>
__global__ __launch_bounds__(Geo<LOG2N>::WAVES * 64, LOG2N <= 9 ? 4 : (LOG2N == 10 ? 2 : 1)) void smile_lld_kernel(
    const float* __restrict__ wav, const int64_t* __restrict__ clip_off, const int64_t* __restrict__ frame_off,
    int64_t total_frames, float* __restrict__ lld, float* __restrict__ cand, float* __restrict__ octave_dbg,
    const Tables<LOG2N>* __restrict__ gtab) {
    using G = Geo<LOG2N>;
    constexpr int NC = G::NC, NB = G::NB, PPL = G::PPL, HALF = G::HALF, CARRY = G::CARRY;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem<LOG2N>& S = *reinterpret_cast<Smem<LOG2N>*>(smem_raw);
    const Tables<LOG2N>& T = S.tab;

    const int clip = blockIdx.y;
    const int64_t s0 = clip_off[clip];
    const int64_t n_samp = clip_off[clip + 1] - s0;
    const int frame = gtab->frame, hop = gtab->hop;
    const int64_t n_fr = n_samp < frame ? 0 : (n_samp - frame) / hop + 1;
    const int64_t f0 = (int64_t)blockIdx.x * G::RUN;
    if (f0 >= n_fr) return;                            // uniform per workgroup
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;

    {   // tables -> LDS
        const float4* g4 = reinterpret_cast<const float4*>(gtab);
        float4* s4 = reinterpret_cast<float4*>(&S.tab);
        for (int i = tid; i < (int)(sizeof(Tables<LOG2N>) / 16); i += G::WAVES * 64) s4[i] = g4[i];
    }
    __syncthreads();

    float* xs = S.x[w];
    v4f* X = reinterpret_cast<v4f*>(xs);
    v2f* XA = reinterpret_cast<v2f*>(xs);              // first half: a (spline ordinates), later S (octave spectrum)
    v2f* XB = XA + HALF;                               // second half: flags / m (spline coefficients), later H (SHS)
    v2f* M = reinterpret_cast<v2f*>(S.mag[w]);
    const float* src = wav + s0;
    const float df = T.df;
    const float inv_frame = 1.0f / (float)frame;

    const int64_t fw = f0 + (int64_t)w * (2 * G::PPW);  // first frame of this wave
    bool have_prev = false;
    v2f prevB[PPL];
    float prevBx = 0.f;                                 // bin NC (kept on lane 63)
#pragma unroll
    for (int i = 0; i < PPL; ++i) prevB[i] = splat(0.f);

    // pass -1 transforms the frame in front of the wave's span (spectral-flux history only: window + FFT + magnitudes,
    // a quarter of a full pass); the waves stay independent of each other
    const int p_first = fw > 0 ? -1 : 0;
#pragma unroll 1
    for (int p = p_first; p < G::PPW; ++p) {
        const bool warm = p < 0;
        const int64_t fA = warm ? fw - 1 : fw + 2 * p;
        if (fA >= n_fr || fw >= n_fr) break;            // wave-uniform
        const bool validB = !warm && (fA + 1 < n_fr);
        const int offB = warm ? 0 : hop;
        const int fl = (int)(fA - f0);                  // local frame index of A within the run

        // ---- stage the pair's samples (coalesced; overlap between pairs and waves is served by L2) ----
        {
            const int span = frame + offB;
            const int64_t sA = fA * hop;
            for (int i = lane; i < span; i += 64) {
                const int64_t si = sA + i;
                xs[i] = (si < n_samp) ? src[si] : 0.0f;
            }
        }
        lds_fence();

        // ---- pre-emphasis, Hamming, frame energies; first radix-4 stage straight from registers ----
        v2f e_rms = splat(0.f), e_int = splat(0.f), zc = splat(0.f);
        {
            constexpr int NBF = NC / 4;
            constexpr int U = (NBF + 63) / 64;
            C2 v[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = lane + 64 * u;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int i0 = 2 * (j + t * NBF), i1 = i0 + 1;
                    v2f y0 = splat(0.f), y1 = splat(0.f);
                    if ((NBF >= 64 || j < NBF) && i0 < frame) {
                        const bool has1 = i1 < frame;
                        const float h0 = T.ham[i0], h1 = T.ham[i1];      // 0 beyond the frame
                        const v2f S0 = {xs[i0], xs[i0 + offB]};
                        const v2f S1 = has1 ? (v2f){xs[i1], xs[i1 + offB]} : splat(0.f);
                        const v2f SM = i0 > 0 ? (v2f){xs[i0 - 1], xs[i0 - 1 + offB]} : splat(0.f);
                        const v2f p0 = i0 > 0 ? (S0 - PREEMPH * SM) : (S0 * (1.0f - PREEMPH));
                        const v2f p1 = S1 - PREEMPH * S0;
                        y0 = p0 * h0;
                        y1 = p1 * h1;
                        e_rms += y0 * y0 + y1 * y1;
                        e_int += h0 * (y0 * y0) + h1 * (y1 * y1);
                        const v2f c0 = S0 * SM, c1 = S1 * S0;
                        zc += (v2f){(i0 > 0 && c0.x < 0.f) ? 1.f : 0.f, (i0 > 0 && c0.y < 0.f) ? 1.f : 0.f};
                        zc += (v2f){(has1 && c1.x < 0.f) ? 1.f : 0.f, (has1 && c1.y < 0.f) ? 1.f : 0.f};
                    }
                    v[u][t] = C2{y0, y1};
                }
                radix4(v[u][0], v[u][1], v[u][2], v[u][3]);
            }
            lds_fence();                                   // every lane has read its samples: the buffer becomes the FFT array
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = lane + 64 * u;
                if (NBF >= 64 || j < NBF) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) stc(X, 4 * j + t, v[u][t]);
                }
            }
            lds_fence();
        }
        fft_rest<NC, 4>(X, T.twr, lane);

        // ---- packed-real unpack -> magnitudes (bin k = lane + 64 q), to the wave's magnitude slot ----
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int k = lane + 64 * q;
            const C2 zk = ldc(X, k);
            const C2 zn = ldc(X, (NC - k) & (NC - 1));
            const v2f ex = 0.5f * (zk.re + zn.re), ey = 0.5f * (zk.im - zn.im);
            const v2f dx = zk.re - zn.re, dy = zk.im + zn.im;
            v2f orr = 0.5f * dy, oi = -0.5f * dx;
            cmul2(orr, oi, T.twr[k]);
            const v2f xr = ex + orr, xi = ey + oi;
            const v2f m2 = xr * xr + xi * xi;
            M[sw(k)] = fsqrt2(m2);
            if (k == 0) {
                const v2f ny = zk.re - zk.im;
                M[sw(NC)] = (v2f){fabsf(ny.x), fabsf(ny.y)};
            }
        }
        lds_fence();

        // ---- consecutive layout: lane owns bins PPL*lane .. PPL*lane + PPL-1, lane 63 also bin NC ----
        const int b0 = PPL * lane;
        const int sb0 = sw(b0);                            // a lane's PPL entries stay contiguous (PPL divides 32)
        v2f m[PPL];
#pragma unroll
        for (int i = 0; i < PPL; ++i) m[i] = M[sb0 + i];
        const v2f mleft = lane > 0 ? M[sw(b0 - 1)] : splat(0.f);
        const v2f mright = M[sw(b0 + PPL)];                    // lane 63: bin NC
        const bool last = lane == 63;
        const v2f mx = last ? mright : splat(0.f);

        if (warm) {
#pragma unroll
            for (int i = 0; i < PPL; ++i) prevB[i] = m[i];
            prevBx = mx.x;
            have_prev = true;
            continue;
        }

        // ---- HTK mel bank: one chunk of one band side per lane (host-built balanced schedule), then per-band sums ----
        {
            v2f* part = reinterpret_cast<v2f*>(S.melbuf[w]);
            v2f* lm = part + 64;
            const int ml = T.mel_lane[lane];
            const int mb = ml & 0xFFF, mlen = (ml >> 12) & 0xFF;
            const bool rising = (ml >> 20) & 1;
            v2f acc = splat(0.f);
            const int iters = gtab->mel_iters;
#pragma unroll 2
            for (int it = 0; it < iters; ++it) {
                if (it < mlen) {
                    const float wt = T.lo_wt[mb + it];
                    acc += (rising ? 1.0f - wt : wt) * M[sw(mb + it)];
                }
            }
            part[lane] = acc;
            lds_fence();
            if (lane < NMEL) {
                const int bd = T.mel_band[lane];
                const int first = bd & 0xFF, n = bd >> 8;
                v2f band = splat(0.f);
                const int mx_n = gtab->mel_max_n;
                for (int k = 0; k < mx_n; ++k)
                    if (k < n) band += part[first + k];
                band = band * HTK_SCALE;
                lm[lane] = (v2f){flog(fmaxf(band.x, MEL_FLOOR)), flog(fmaxf(band.y, MEL_FLOOR))};
            }
            lds_fence();
            // DCT-II + lifter: lane 4k + part sums 7 mel channels of cepstral coefficient k + 1
            const int kk = lane >> 2, prt = lane & 3;
            v2f dsum = splat(0.f);
            if (kk < NMFCC) {
#pragma unroll
                for (int jj = 0; jj < 7; ++jj) {
                    const int j = 7 * prt + jj;
                    if (j < NMEL) dsum += T.dct[kk * NMEL + j] * lm[j];
                }
            }
            dsum += (v2f){dpp_f32<0xB1>(dsum.x), dpp_f32<0xB1>(dsum.y)};          // quad_perm [1,0,3,2]
            dsum += (v2f){dpp_f32<0x4E>(dsum.x), dpp_f32<0x4E>(dsum.y)};          // quad_perm [2,3,0,1]
            if (kk < NMFCC && prt == 0) *reinterpret_cast<v2f*>(&S.out[1 + kk][fl]) = dsum;
        }

        // ---- cSpectral on the power spectrum ----
        v2f s_p = splat(0.f), s_fp = splat(0.f), s_b1 = splat(0.f), s_b2 = splat(0.f), s_fl = splat(0.f),
            s_sh = splat(0.f), s_m = splat(0.f), s_lg = splat(0.f), s_pk = splat(0.f);
        const bool flux_a = have_prev;
        const float b1lo = gtab->band1_lo, b1hi = gtab->band1_hi, b2lo = gtab->band2_lo, b2hi = gtab->band2_hi;
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const int b = b0 + i;
            const float fq = b * df;
            const v2f p = m[i] * m[i];
            s_p += p;
            s_fp += p * fq;
            if (fq >= b1lo && fq <= b1hi) s_b1 += p;
            if (fq >= b2lo && fq <= b2hi) s_b2 += p;
            const v2f dm = {flux_a ? m[i].x - prevB[i].y : 0.f, m[i].y - m[i].x};   // B's history is A
            s_fl += dm * dm;
            s_sh += p * T.sharp[b];
            s_m += m[i];
            s_lg += flog2_2(vmax2(p, splat(1e-30f)));
            if (b >= 1) {
                const v2f ml_ = i > 0 ? m[i - 1] : mleft;
                const v2f mr_ = i < PPL - 1 ? m[i + 1] : mright;
                s_pk += vmax2(m[i] - 0.5f * (ml_ + mr_), splat(0.f));      // prominence over the neighbours' mean
            }
        }
        const v2f px = mx * mx;                                                // bin NC (lane 63 only, 0 elsewhere)
        if (last) {
            const float fq = NC * df;
            s_fp += px * fq;
            if (fq >= b1lo && fq <= b1hi) s_b1 += px;
            if (fq >= b2lo && fq <= b2hi) s_b2 += px;
            const v2f dm = {flux_a ? mx.x - prevBx : 0.f, mx.y - mx.x};
            s_fl += dm * dm;
            s_sh += px * T.sharp[NC];
            s_m += mx;
            s_lg += flog2_2(vmax2(px, splat(1e-30f)));
        }
        {   // window sums of the pair (a fourth slot of the group is free)
            v2f dummy = splat(0.f);
            wave_sum4x2(e_rms, e_int, zc, dummy);
            if (lane == 0) {
                auto put0 = [&](int row, v2f val) { *reinterpret_cast<v2f*>(&S.out[local_row(row)][fl]) = val; };
                const v2f inten = e_int * (1.0e6f / gtab->ham_sum);
                put0(0, fsqrt2(e_rms * inv_frame));
                put0(13, zc * inv_frame);
                put0(16, inten);
                put0(17, fexp2_2(0.3f * flog2_2(inten)));                   // inten^0.3 (0 -> 0)
            }
        }
        // inclusive scan of the per-lane power (bins 0..NC-1); the total adds bin NC (held by lane 63)
        const v2f incl = {wave_scan_incl(s_p.x), wave_scan_incl(s_p.y)};
        const v2f tot = (v2f){readlane_f32(incl.x, 63) + readlane_f32(px.x, 63), readlane_f32(incl.y, 63) + readlane_f32(px.y, 63)};
        const v2f excl = incl - s_p;
        wave_sum4x2(s_fp, s_b1, s_b2, s_fl);
        wave_sum4x2(s_sh, s_m, s_lg, s_pk);
        const v2f tot_fp = s_fp, band1 = s_b1, band2 = s_b2, flsum = s_fl, sharp = s_sh, msum = s_m, pksum = s_pk;
        const v2f lgsum = s_lg * 0.6931471805599453f;                          // sum of natural logs
        const v2f safe = {tot.x > 0.f ? tot.x : 1.0f, tot.y > 0.f ? tot.y : 1.0f};
        const v2f inv = frcp2(safe);
        const v2f cen = tot_fp * inv;
        // roll-off: first bin whose inclusive cumulative power reaches p * total
        float ro[8];
        {
            const float pr[4] = {0.25f, 0.50f, 0.75f, 0.90f};
            v2f run = excl;
            v2f cs[PPL];
#pragma unroll
            for (int i = 0; i < PPL; ++i) { run += m[i] * m[i]; cs[i] = run; }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const v2f thr = pr[t] * tot;
                int cA = NC, cB = NC;
#pragma unroll
                for (int i = PPL - 1; i >= 0; --i) {
                    if (cs[i].x >= thr.x) cA = b0 + i;
                    if (cs[i].y >= thr.y) cB = b0 + i;
                }
                ro[2 * t] = (float)cA;                                          // bin indices are exact in float
                ro[2 * t + 1] = (float)cB;
            }
            wave_reduce8<true>(ro);
        }
        // second pass: central moments + entropy
        v2f s_e = splat(0.f), s_v = splat(0.f), s_s = splat(0.f), s_k = splat(0.f);
#pragma unroll
        for (int i = 0; i <= PPL; ++i) {
            if (i == PPL && !last) break;
            const float fq = (i < PPL ? b0 + i : NC) * df;
            const v2f mi_ = i < PPL ? m[i] : mx;
            const v2f prb = mi_ * mi_ * inv;
            const v2f d = splat(fq) - cen;
            const v2f lg = flog2_2(vmax2(prb, splat(1e-37f)));
            s_e += (v2f){prb.x > 0.f ? prb.x * lg.x : 0.f, prb.y > 0.f ? prb.y * lg.y : 0.f};
            const v2f d2 = d * d;
            s_v += d2 * prb;
            s_s += d2 * d * prb;
            s_k += d2 * d2 * prb;
        }
        wave_sum4x2(s_e, s_v, s_s, s_k);
        // the three window sums of this pair travel with the sub-harmonic sum below (one more group of four)
        if (lane == 0) {
            const v2f var = s_v;
            const v2f vs = {var.x > 0.f ? var.x : 1.0f, var.y > 0.f ? var.y : 1.0f};
            const v2f ivs = frcp2(vs);
            auto put = [&](int row, v2f val) { *reinterpret_cast<v2f*>(&S.out[local_row(row)][fl]) = val; };
            put(22, band1);
            put(23, band2);
            put(24, (v2f){ro[0], ro[1]} * df);
            put(25, (v2f){ro[2], ro[3]} * df);
            put(26, (v2f){ro[4], ro[5]} * df);
            put(27, (v2f){ro[6], ro[7]} * df);
            put(28, fsqrt2(flsum * (1.0f / NB)));                           // the clip's first frame: history = itself -> 0
            put(29, cen);
            put(30, -s_e);
            put(31, var);
            put(32, s_s * ivs * fsqrt2(ivs));
            put(33, s_k * ivs * ivs);
            put(34, ((float)NB * tot_fp - gtab->slope_sf * tot) * (1.0f / gtab->slope_den));
            put(35, sharp * inv);
            put(36, pksum * frcp2((v2f){msum.x > 0.f ? msum.x : 1.0f, msum.y > 0.f ? msum.y : 1.0f}));
            put(37, fexp2_2(s_lg * (1.0f / NB)) * frcp2(vmax2(tot * (1.0f / NB), splat(1e-30f))));
        }
#pragma unroll
        for (int i = 0; i < PPL; ++i) prevB[i] = m[i];
        prevBx = mx.y;
        have_prev = true;
        (void)lgsum;

        // ---- cSpecScale: peak enhancement + smoothing on the linear spectrum ----
        // flags word per bin: bit 0 / 1 = "local maximum" in frame A / B; first / last / count from wave ballots (SALU)
        unsigned* FL = reinterpret_cast<unsigned*>(XB);
        int gfirstA = 0x7fffffff, gfirstB = 0x7fffffff, glastA = -1, glastB = -1, ncA = 0, ncB = 0;
#pragma unroll
        for (int i = 0; i <= PPL; ++i) {
            const int b = i < PPL ? b0 + i : NC;
            const v2f me = i < PPL ? m[i] : mx;
            const v2f ml_ = i == 0 ? mleft : m[i - 1];
            const v2f mr_ = i < PPL - 1 ? m[i + 1] : (i == PPL - 1 ? mright : splat(0.f));
            bool fa, fb;
            if (i == PPL) { fa = last && me.x > ml_.x; fb = last && me.y > ml_.y; }
            else if (b == 0) { fa = me.x > mr_.x; fb = me.y > mr_.y; }
            else { fa = me.x > ml_.x && me.x >= mr_.x; fb = me.y > ml_.y && me.y >= mr_.y; }
            if (i < PPL || last) FL[b] = (fa ? 1u : 0u) | (fb ? 2u : 0u);
            const unsigned long long ka = __ballot(fa), kb = __ballot(fb);
            const int stride = i < PPL ? PPL : 0, base = i < PPL ? i : NC;
            if (ka) {
                gfirstA = min(gfirstA, stride * (__ffsll((long long)ka) - 1) + base);
                glastA = max(glastA, stride * (63 - __clzll((long long)ka)) + base);
                ncA += __popcll(ka);
            }
            if (kb) {
                gfirstB = min(gfirstB, stride * (__ffsll((long long)kb) - 1) + base);
                glastB = max(glastB, stride * (63 - __clzll((long long)kb)) + base);
                ncB += __popcll(kb);
            }
        }
        lds_fence();
        // window of flags for bins b0-3 .. b0+PPL+2
        unsigned fw_[PPL + 6];
#pragma unroll
        for (int o = 0; o < PPL + 6; ++o) {
            const int b = b0 - 3 + o;
            fw_[o] = (b >= 0 && b <= NC) ? FL[b] : 0u;
        }
        // enhanced value of bin b0 - 1 + o, o = 0 .. PPL+1 (own bins and one neighbour on each side)
        v2f en[PPL + 2];
#pragma unroll
        for (int o = 0; o < PPL + 2; ++o) {
            const int b = b0 - 1 + o;
            const unsigned near = fw_[o] | fw_[o + 1] | fw_[o + 2] | fw_[o + 3] | fw_[o + 4];   // bins b-2 .. b+2
            const v2f val = o == 0 ? mleft : (o <= PPL ? m[o - 1] : mright);
            const bool zA = !(near & 1u) && (ncA == 1 || (ncA >= 2 && b > gfirstA && b < glastA));
            const bool zB = !(near & 2u) && (ncB == 1 || (ncB >= 2 && b > gfirstB && b < glastB));
            en[o] = (b >= 0 && b <= NC) ? (v2f){zA ? 0.f : val.x, zB ? 0.f : val.y} : splat(0.f);
        }
        lds_fence();                                         // flags consumed: the second half is free again
        // smoothing (1, 2, 1) / 4, zero left of bin 0, bin NC untouched; a -> first half
        v2f a[PPL];
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            a[i] = 0.25f * (en[i] + 2.0f * en[i + 1] + en[i + 2]);
            XA[sb0 + i] = a[i];
        }
        if (last) XA[sw(NC)] = en[PPL + 1];                      // bin NC keeps its enhanced value
        lds_fence();
        // ---- natural cubic spline through the bins: m_{b-1} + 4 m_b + m_{b+1} = a_{b-1} - 2 a_b + a_{b+1} ----
        {
            const v2f aleft = lane > 0 ? XA[sw(b0 - 1)] : splat(0.f);
            const v2f aright = XA[sw(b0 + PPL)];
            float g[PPL];
            v2f d[PPL];
            // forward elimination, local part (carry 0), then the carry of up to CARRY lanes
            v2f run = splat(0.f);
            float qf[PPL];
            float q = 1.0f;
#pragma unroll
            for (int i = 0; i < PPL; ++i) {
                g[i] = T.sp_g[b0 + i];
                const v2f al = i == 0 ? aleft : a[i - 1];
                const v2f ar = i == PPL - 1 ? aright : a[i + 1];
                const v2f r = al - 2.0f * a[i] + ar;
                run = g[i] * (r - run);
                d[i] = run;
                q *= -g[i];
                qf[i] = q;
            }
            {
                v2f e = run, carry = splat(0.f);
#pragma unroll
                for (int dd = 1; dd <= CARRY; ++dd) {
                    e = (v2f){wave_shr1(e.x), wave_shr1(e.y)};
                    carry += (dd == 1 ? 1.0f : T.sp_cf[dd - 2][lane]) * e;
                }
#pragma unroll
                for (int i = 0; i < PPL; ++i) d[i] += qf[i] * carry;
            }
            // back substitution x_b = dp_b - g_b x_{b+1}
            v2f x[PPL];
            float qb[PPL];
            run = splat(0.f);
            q = 1.0f;
#pragma unroll
            for (int i = PPL - 1; i >= 0; --i) {
                run = d[i] - g[i] * run;
                x[i] = run;
                q *= -g[i];
                qb[i] = q;
            }
            {
                v2f e = run, carry = splat(0.f);
#pragma unroll
                for (int dd = 1; dd <= CARRY; ++dd) {
                    e = (v2f){wave_shl1(e.x), wave_shl1(e.y)};
                    carry += (dd == 1 ? 1.0f : T.sp_cb[dd - 2][lane]) * e;
                }
#pragma unroll
                for (int i = 0; i < PPL; ++i) x[i] += qb[i] * carry;
            }
#pragma unroll
            for (int i = 0; i < PPL; ++i) XB[sb0 + i] = x[i];
            if (last) XB[sw(NC)] = splat(0.f);
        }
        lds_fence();
        // ---- octave-scale targets i = b0 .. b0+PPL-1 (+ NC on lane 63): spline value, clip, auditory weighting ----
        v2f sv[PPL + 1];
#pragma unroll
        for (int i = 0; i <= PPL; ++i) {
            sv[i] = splat(0.f);
            if (i == PPL && !last) break;
            const int ti = i < PPL ? b0 + i : NC;
            const int k = T.klo[ti];
            const float bb = T.tb[ti], aa = 1.0f - bb;
            const float ca = aa * aa * aa - aa, cb = bb * bb * bb - bb;
            const v2f y = aa * XA[sw(k)] + bb * XA[sw(k + 1)] + ca * XB[sw(k)] + cb * XB[sw(k + 1)];
            sv[i] = vmax2(y, splat(0.f)) * T.audw[ti];
        }
        lds_fence();                                         // all reads of a / m done: reuse the halves for S / H
#pragma unroll
        for (int i = 0; i < PPL; ++i) XA[sb0 + i] = sv[i];
        if (last) { XA[sw(NC)] = sv[PPL]; XA[sw(NC + 1)] = splat(0.f); }   // entry NC+1 = 0: target of out-of-range shifts
        if (octave_dbg) {
            const int64_t fg = frame_off[clip] + fA;
#pragma unroll
            for (int i = 0; i <= PPL; ++i) {
                if (i == PPL && !last) break;
                const int ti = i < PPL ? b0 + i : NC;
                octave_dbg[fg * NB + ti] = sv[i].x;
                if (validB) octave_dbg[(fg + 1) * NB + ti] = sv[i].y;
            }
        }
        lds_fence();
        // ---- cPitchShs: sub-harmonic summation H[i] = sum_h 0.85^(h-1) S[i + shift_h] ----
        v2f hv[PPL + 1];
        v2f hsum = splat(0.f);
#pragma unroll
        for (int i = 0; i <= PPL; ++i) {
            hv[i] = splat(0.f);
            if (i == PPL && !last) break;
            const int ti = i < PPL ? b0 + i : NC;
            v2f acc = splat(0.f);
#pragma unroll
            for (int h = 0; h < NHARM; ++h) {
                const int src_i = min(ti + gtab->shs_shift[h], NC + 1);
                acc += gtab->shs_w[h] * XA[sw(src_i)];
            }
            hv[i] = acc;
            hsum += acc;
        }
        const v2f hmean = wave_sum2(hsum) * (1.0f / NB);
#pragma unroll
        for (int i = 0; i < PPL; ++i) XB[sb0 + i] = hv[i];
        if (last) XB[sw(NC)] = hv[PPL];
        lds_fence();
        {
            // local maxima with parabolic refinement, both frames at once; a peak's sort key carries its score in the
            // high bits and (MASK - index) in the low ones: larger key = higher score, ties / near-ties to the lower index
            constexpr unsigned MASK = (1u << (LOG2N - 1)) - 1u;
            constexpr int MAXPK = PPL >= 2 ? PPL / 2 : 1;                    // peaks are never adjacent
            const v2f hleft = lane > 0 ? XB[sw(b0 - 1)] : splat(0.f);
            const v2f hright = XB[sw(b0 + PPL)];
            unsigned kA[MAXPK], kB[MAXPK];
#pragma unroll
            for (int k = 0; k < MAXPK; ++k) { kA[k] = 0u; kB[k] = 0u; }
            const float fmin_l2 = gtab->fmin_l2, dl2 = gtab->dl2;
#pragma unroll
            for (int i = 0; i < PPL; ++i) {
                const int ti = b0 + i;
                const v2f y1 = i == 0 ? hleft : hv[i - 1];
                const v2f y2 = hv[i];
                const v2f y3 = i == PPL - 1 ? hright : hv[i + 1];
                const v2f den = y1 - 2.0f * y2 + y3;
                const v2f rden = frcp2(den);
                const v2f dif = y1 - y3;
                const v2f sc = y2 - 0.125f * dif * dif * rden;
                const v2f fq = fexp2_2(splat(fmin_l2) + (splat((float)ti) + 0.5f * dif * rden) * dl2);
                const bool okA = ti >= 1 && y2.x > y1.x && y2.x >= y3.x && fq.x >= 52.0f && fq.x <= 620.0f && sc.x > 0.f;
                const bool okB = ti >= 1 && y2.y > y1.y && y2.y >= y3.y && fq.y >= 52.0f && fq.y <= 620.0f && sc.y > 0.f;
                unsigned ka = okA ? ((__float_as_uint(sc.x) & ~MASK) | (MASK - (unsigned)ti)) : 0u;
                unsigned kb = okB ? ((__float_as_uint(sc.y) & ~MASK) | (MASK - (unsigned)ti)) : 0u;
#pragma unroll
                for (int k = 0; k < MAXPK; ++k) {                             // sorted insertion, largest first
                    const unsigned ta = max(kA[k], ka), tb = max(kB[k], kb);
                    ka = min(kA[k], ka); kb = min(kB[k], kb);
                    kA[k] = ta; kB[k] = tb;
                }
            }
            unsigned wA = 0u, wB = 0u;                                       // lane r keeps the key of slot r
#pragma unroll 1
            for (int r = 0; r < NCAND; ++r) {
                const unsigned ma = wave_max_u32(kA[0]), mb_ = wave_max_u32(kB[0]);
                if (!(ma | mb_)) break;                                       // wave-uniform
                if (lane == r) { wA = ma; wB = mb_; }
                if (kA[0] == ma && ma) {
#pragma unroll
                    for (int k = 0; k + 1 < MAXPK; ++k) kA[k] = kA[k + 1];
                    kA[MAXPK - 1] = 0u;
                }
                if (kB[0] == mb_ && mb_) {
#pragma unroll
                    for (int k = 0; k + 1 < MAXPK; ++k) kB[k] = kB[k + 1];
                    kB[MAXPK - 1] = 0u;
                }
            }
            if (lane < NCAND) {
                const int64_t fg = frame_off[clip] + fA;
                const float* Hf = reinterpret_cast<const float*>(XB);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && !validB) break;
                    const unsigned key = h ? wB : wA;
                    float fqo = 0.f, vo = 0.f;
                    if (key) {
                        const int ti = (int)(MASK - (key & MASK));
                        const float y1 = Hf[2 * sw(ti - 1) + h], y2 = Hf[2 * sw(ti) + h], y3 = Hf[2 * sw(ti + 1) + h];
                        const float rden = frcp(y1 - 2.0f * y2 + y3);
                        const float dif = y1 - y3;
                        const float sc = y2 - 0.125f * dif * dif * rden;
                        fqo = fexp2(fmin_l2 + ((float)ti + 0.5f * dif * rden) * dl2);
                        vo = fmaxf(0.f, 1.0f - (h ? hmean.y : hmean.x) * frcp(sc));
                    }
                    reinterpret_cast<float2*>(cand)[(fg + h) * NCAND + lane] = make_float2(fqo, vo);
                }
            }
        }
        lds_fence();
    }
    __syncthreads();

    // ---- coalesced contour-major store of the run ----
    const int64_t fbase = frame_off[clip] + f0;
    const int nvalid = (int)min((int64_t)G::RUN, n_fr - f0);
    for (int idx = tid; idx < NLOCAL * G::RUN; idx += G::WAVES * 64) {
        const int r = idx / G::RUN, t = idx % G::RUN;
        if (t >= nvalid) continue;
        lld[(int64_t)lld_of_local(r) * total_frames + fbase + t] = S.out[r][t];
    }
}

template <int LOG2N>
static int launch(const float* wav, const int64_t* clip_off, const int64_t* frame_off, int n_clips,
                  int64_t max_clip_frames, int64_t total_frames, int fs, int frame, int hop, float* lld, float* cand,
                  float* octave_dbg, hipStream_t s) {
    using G = Geo<LOG2N>;
    const Tables<LOG2N>* tab = nullptr;
    int rc = get_tables<LOG2N>(fs, frame, hop, &tab);
    if (rc != RSAF_OK) return rc;
    const int64_t runs = (max_clip_frames + G::RUN - 1) / G::RUN;
    RSAF_CHECK_ARG(runs <= 0x7fffffffLL, "clip too long");
    static bool attr_set[64] = {false};
    int dev = 0;
    RSAF_CHECK_HIP(hipGetDevice(&dev));
    RSAF_CHECK_ARG(dev >= 0 && dev < 64, "device index out of range");
    if (!attr_set[dev]) {
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)smile_lld_kernel<LOG2N>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem<LOG2N>)));
        attr_set[dev] = true;
    }
    // algorithmic bytes: every sample read once (4 B) + the LLD rows written (38 * 4 B per frame)
    ProfScope prof("smile_lld", s, 0.0, 0.0);
    dim3 grid((unsigned)runs, (unsigned)n_clips);
    hipLaunchKernelGGL(smile_lld_kernel<LOG2N>, grid, dim3(G::WAVES * 64), sizeof(Smem<LOG2N>), s, wav, clip_off,
                       frame_off, total_frames, lld, cand, octave_dbg, tab);
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
                         int n_clips, int64_t max_clip_frames, int64_t total_frames, int sample_rate, float* lld,
                         float* cand, float* octave_spectrum, rsaf_stream_t stream) {
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
