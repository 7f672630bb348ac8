// openSMILE-style low-level descriptors for gfx950 (wave64).
//
// One fused kernel for Androids.conf:73-139 and :258-280 of the reference
// (cFramer -> cVectorPreemphasis -> cWindower -> cTransformFFT -> cFFTmagphase ->
//  {cMelspec -> cMfcc, cEnergy, cMZcr, cIntensity, cSpectral}); the reference runs that chain by
// spawning SMILExtract once per file (src/opensmile_extractor.py:62-75).
//
// Mapping: a 256-thread workgroup owns a run of 32 consecutive frames of one clip.  The run's
// samples (31*160+400, plus one hop of history for the spectral-flux warm-up frame) are staged
// once in LDS with coalesced loads, so the 2.5x frame overlap never re-reads HBM.  Each wave then
// walks 8 consecutive frames: packed-real 512-point FFT as a 256-point complex Stockham radix-4
// FFT (4 stages, 4 points per lane, per-wave LDS ping-pong, twiddles from LDS), magnitudes to LDS,
// HTK mel bank as 27 segment sums, log + DCT/lifter from LDS, spectral descriptors as wave
// reductions (DPP).  The 32 built LLD rows are buffered per run in LDS and written contour-major
// with 128-byte coalesced rows.
//
// Semantics are those of oracle/smile_oracle.py (the CPU restatement), parity unpinned.
#include <cmath>
#include <mutex>
#include <vector>

#include "rsaf_common.h"

namespace rsaf {
namespace smile {

constexpr int FRAME = RSAF_SMILE_FRAME;
constexpr int HOP = RSAF_SMILE_HOP;
constexpr int NFFT = 512;
constexpr int NBINS = 257;
constexpr int NMEL = 26;
constexpr int NMFCC = 12;
constexpr int NLLD = RSAF_SMILE_NLLD;
constexpr float PREEMPH = 0.97f;
constexpr float HTK_SCALE = 32767.0f;
constexpr float MEL_FLOOR = 1.0f;
constexpr float DF = 16000.0f / NFFT;

constexpr int RUN = 32;                               // frames per workgroup
constexpr int FPW = RUN / 4;                          // frames per wave
constexpr int NSAMP = (RUN - 1) * HOP + FRAME + HOP;  // 5520 incl. one hop of history

// ---- constant tables (host-computed in double, uploaded once per device) -----------------
struct Tables {
    float ham[FRAME];
    float2 tw256[256];     // exp(-2 pi i m / 256)
    float2 tw512[256];     // exp(-2 pi i k / 512)
    float lo_wt[260];      // HTK lower-channel weight per bin (0 where unused)
    int seg_start[32];     // bins with lo_chan == c are [seg_start[c], seg_start[c+1]), c = 0..26
    float dct[NMFCC * NMEL];  // DCT-II rows 1..12 with the lifter folded in
    float sharp[260];      // bark(f) * g(bark) per bin
    float ham_sum;
    float pad[3];
};
static_assert(sizeof(Tables) % 16 == 0, "Tables must be float4-copyable");

static double mel_d(double f) { return 2595.0 * std::log10(1.0 + f / 700.0); }

static void build_tables(Tables& t) {
    std::memset(&t, 0, sizeof(t));
    double hs = 0;
    for (int i = 0; i < FRAME; ++i) {
        double w = 0.54 - 0.46 * std::cos(2.0 * M_PI * i / (FRAME - 1));
        t.ham[i] = (float)w;
        hs += w;
    }
    t.ham_sum = (float)hs;
    for (int m = 0; m < 256; ++m) {
        t.tw256[m] = make_float2((float)std::cos(2.0 * M_PI * m / 256), (float)-std::sin(2.0 * M_PI * m / 256));
        t.tw512[m] = make_float2((float)std::cos(2.0 * M_PI * m / 512), (float)-std::sin(2.0 * M_PI * m / 512));
    }
    // HTK filterbank: centre frequencies equally spaced on the mel scale between 20 and 8000 Hz
    const double lo = mel_d(20.0), hi = mel_d(8000.0);
    double cf[NMEL + 2];
    for (int c = 0; c < NMEL + 2; ++c) cf[c] = lo + (hi - lo) * c / (NMEL + 1);
    int lo_chan[NBINS];
    for (int b = 0; b < NBINS; ++b) {
        double f = b * (16000.0 / NFFT);
        lo_chan[b] = -1;
        if (f < 20.0 || f > 8000.0) continue;
        double m = mel_d(f);
        int c = 0;
        while (c < NMEL && cf[c + 1] <= m) ++c;   // cf[c] <= m < cf[c+1], capped at NMEL
        lo_chan[b] = c;
        t.lo_wt[b] = (float)((cf[c + 1] - m) / (cf[c + 1] - cf[c]));
    }
    int b = 0;
    while (b < NBINS && lo_chan[b] < 0) ++b;
    for (int c = 0; c <= NMEL + 1; ++c) {
        while (b < NBINS && lo_chan[b] >= 0 && lo_chan[b] < c) ++b;
        t.seg_start[c] = b;
    }
    for (int k = 1; k <= NMFCC; ++k) {
        double lift = 1.0 + 11.0 * std::sin(M_PI * k / 22.0);
        for (int j = 1; j <= NMEL; ++j)
            t.dct[(k - 1) * NMEL + (j - 1)] =
                (float)(std::sqrt(2.0 / NMEL) * std::cos(M_PI * k * (j - 0.5) / NMEL) * lift);
    }
    for (int bb = 0; bb < NBINS; ++bb) {
        double f = bb * (16000.0 / NFFT);
        double z = 13.0 * std::atan(0.00076 * f) + 3.5 * std::atan((f / 7500.0) * (f / 7500.0));
        double g = z < 14.0 ? 1.0 : 0.066 * std::exp(0.171 * z);
        t.sharp[bb] = (float)(z * g);
    }
}

static std::mutex g_mu;
static Tables* g_dev_tables[64] = {nullptr};

int get_tables(const Tables** out) {
    int dev = 0;
    RSAF_CHECK_HIP(hipGetDevice(&dev));
    RSAF_CHECK_ARG(dev >= 0 && dev < 64, "device index out of range");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_dev_tables[dev]) {
        std::vector<Tables> h(1);
        build_tables(h[0]);
        Tables* d = nullptr;
        RSAF_CHECK_HIP(hipMalloc(&d, sizeof(Tables)));
        RSAF_CHECK_HIP(hipMemcpy(d, h.data(), sizeof(Tables), hipMemcpyHostToDevice));
        g_dev_tables[dev] = d;
    }
    *out = g_dev_tables[dev];
    return RSAF_OK;
}

// ---- device code -----------------------------------------------------------------------------
struct __attribute__((aligned(16))) Smem {
    Tables tab;
    float samp[NSAMP];
    float2 fa[4][256];       // per-wave FFT ping
    float2 fb[4][256];       // per-wave FFT pong
    float mag[4][2][260];    // per-wave magnitude spectra (current / previous frame)
    float logmel[4][32];
    float out[NLLD][RUN];    // LLD rows of this run
};

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ void radix4(float2& v0, float2& v1, float2& v2, float2& v3) {
    float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
    float2 a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
    float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
    float2 t = make_float2(v1.x - v3.x, v1.y - v3.y);
    float2 a3 = make_float2(t.y, -t.x);   // -i * t
    v0 = make_float2(a0.x + a2.x, a0.y + a2.y);
    v1 = make_float2(a1.x + a3.x, a1.y + a3.y);
    v2 = make_float2(a0.x - a2.x, a0.y - a2.y);
    v3 = make_float2(a1.x - a3.x, a1.y - a3.y);
}

template <int NS>
__device__ __forceinline__ void stockham_stage(const float2* __restrict__ src, float2* __restrict__ dst,
                                               const float2* __restrict__ tw, int j) {
    float2 v0 = src[j], v1 = src[j + 64], v2 = src[j + 128], v3 = src[j + 192];
    const int k = j & (NS - 1);
    const int m = k * (64 / NS);
    v1 = cmul(v1, tw[m]);
    v2 = cmul(v2, tw[2 * m]);
    v3 = cmul(v3, tw[3 * m]);
    radix4(v0, v1, v2, v3);
    const int j0 = ((j - k) << 2) + k;
    dst[j0] = v0;
    dst[j0 + NS] = v1;
    dst[j0 + 2 * NS] = v2;
    dst[j0 + 3 * NS] = v3;
}

__global__ __launch_bounds__(256) void smile_lld_kernel(const float* __restrict__ wav,
                                                        const int64_t* __restrict__ clip_off,
                                                        const int64_t* __restrict__ frame_off,
                                                        int64_t total_frames,
                                                        float* __restrict__ lld,
                                                        const Tables* __restrict__ gtab) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem& S = *reinterpret_cast<Smem*>(smem_raw);

    const int clip = blockIdx.y;
    const int64_t s0 = clip_off[clip];
    const int64_t n_samp = clip_off[clip + 1] - s0;
    const int64_t n_fr = n_samp < FRAME ? 0 : (n_samp - FRAME) / HOP + 1;
    const int64_t f0 = (int64_t)blockIdx.x * RUN;
    if (f0 >= n_fr) return;                       // uniform per workgroup
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;

    // ---- stage tables + samples (coalesced) ----
    {
        const float4* g4 = reinterpret_cast<const float4*>(gtab);
        float4* s4 = reinterpret_cast<float4*>(&S.tab);
        for (int i = tid; i < (int)(sizeof(Tables) / 16); i += 256) s4[i] = g4[i];
        const int64_t base = f0 * HOP - HOP;      // sample index of S.samp[0] within the clip
        const float* src = wav + s0;
        for (int i = tid; i < NSAMP; i += 256) {
            const int64_t si = base + i;
            S.samp[i] = (si >= 0 && si < n_samp) ? src[si] : 0.0f;
        }
    }
    __syncthreads();

    const float2* tw = S.tab.tw256;
    float2* FA = S.fa[w];
    float2* FB = S.fb[w];
    int cur = 0;
    bool have_prev = false;

    for (int ff = -1; ff < FPW; ++ff) {
        const int fl = w * FPW + ff;              // frame index within the run (-1 = warm-up)
        const int64_t f = f0 + fl;
        if (f < 0) continue;                      // clip's first frame has no predecessor
        if (ff >= 0 && f >= n_fr) break;
        const bool warm = ff < 0;
        const float* x = &S.samp[(fl + 1) * HOP]; // +1: one hop of history in front

        // ---- pre-emphasis, Hamming, frame energies, first Stockham stage from registers ----
        float2 v[4];
        float e_rms = 0.f, e_int = 0.f, zc = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 2 * lane + 128 * r;
            float y0 = 0.f, y1 = 0.f;
            if (i < FRAME) {
                const float xm = (i > 0) ? x[i - 1] : 0.0f;
                const float xa = x[i], xb = x[i + 1];
                const float p0 = (i > 0) ? (xa - PREEMPH * xm) : (xa * (1.0f - PREEMPH));
                const float p1 = xb - PREEMPH * xa;
                const float h0 = S.tab.ham[i], h1 = S.tab.ham[i + 1];
                y0 = p0 * h0;
                y1 = p1 * h1;
                e_rms += y0 * y0 + y1 * y1;
                e_int += h0 * y0 * y0 + h1 * y1 * y1;
                zc += ((i > 0 && xa * xm < 0.0f) ? 1.0f : 0.0f) + ((xb * xa < 0.0f) ? 1.0f : 0.0f);
            }
            v[r] = make_float2(y0, y1);
        }
        radix4(v[0], v[1], v[2], v[3]);           // stage Ns=1 (twiddles are 1)
        {
            float4* d4 = reinterpret_cast<float4*>(FA);
            d4[2 * lane] = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
            d4[2 * lane + 1] = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
        }
        lds_fence();
        stockham_stage<4>(FA, FB, tw, lane);
        lds_fence();
        stockham_stage<16>(FB, FA, tw, lane);
        lds_fence();
        {   // last stage (Ns = 64): outputs stay in registers, a copy goes to FB for the mirror reads
            float2 v0 = FA[lane], v1 = FA[lane + 64], v2 = FA[lane + 128], v3 = FA[lane + 192];
            v1 = cmul(v1, tw[lane]);
            v2 = cmul(v2, tw[2 * lane]);
            v3 = cmul(v3, tw[3 * lane]);
            radix4(v0, v1, v2, v3);
            v[0] = v0; v[1] = v1; v[2] = v2; v[3] = v3;
            FB[lane] = v0; FB[lane + 64] = v1; FB[lane + 128] = v2; FB[lane + 192] = v3;
        }
        lds_fence();
        float* M = S.mag[w][cur];
        const float* Mp = S.mag[w][cur ^ 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = lane + 64 * q;
            const float2 zk = v[q];
            const float2 zn = FB[(256 - k) & 255];
            const float ex = 0.5f * (zk.x + zn.x), ey = 0.5f * (zk.y - zn.y);
            const float dx = zk.x - zn.x, dy = zk.y + zn.y;
            const float2 o = make_float2(0.5f * dy, -0.5f * dx);
            const float2 wo = cmul(S.tab.tw512[k], o);
            const float xr = ex + wo.x, xi = ey + wo.y;
            M[k] = sqrtf(xr * xr + xi * xi);
            if (k == 0) M[256] = fabsf(zk.x - zk.y);
        }
        lds_fence();
        if (warm) { cur ^= 1; have_prev = true; continue; }

        // ---- frame energies ----
        const float rms = sqrtf(wave_sum(e_rms) / FRAME);
        const float inten = wave_sum(e_int) / S.tab.ham_sum * 1.0e6f;
        const float zcr = wave_sum(zc) / FRAME;

        // ---- HTK mel bank as segment sums, log, DCT ----
        {
            float sa = 0.f, sb = 0.f;
            if (lane <= NMEL) {
                const int b0 = S.tab.seg_start[lane], b1 = S.tab.seg_start[lane + 1];
                for (int b = b0; b < b1; ++b) {
                    const float m = M[b];
                    const float a = S.tab.lo_wt[b] * m;
                    sa += a;
                    sb += m - a;
                }
            }
            const float sb_prev = __shfl_up(sb, 1, 64);
            // lane c (1..26) holds band c
            const float band = (sa + sb_prev) * HTK_SCALE;
            if (lane >= 1 && lane <= NMEL) S.logmel[w][lane - 1] = logf(fmaxf(band, MEL_FLOOR));
            lds_fence();
            if (lane < NMFCC) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < NMEL; ++j) acc += S.logmel[w][j] * S.tab.dct[lane * NMEL + j];
                S.out[1 + lane][fl] = acc;
            }
        }

        // ---- cSpectral on the power spectrum: lane owns bins 4*lane .. 4*lane+3 (+ bin 256 on lane 0)
        float m4[4], p4[4];
        {
            const float4 mm = reinterpret_cast<const float4*>(M)[lane];
            m4[0] = mm.x; m4[1] = mm.y; m4[2] = mm.z; m4[3] = mm.w;
        }
        const float m256 = M[256];
        const float p256 = m256 * m256;
        float s_p = 0.f, s_fp = 0.f, s_b1 = 0.f, s_b2 = 0.f, s_fl = 0.f, s_sh = 0.f, s_m = 0.f, s_lg = 0.f,
              s_pk = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = 4 * lane + i;
            const float fq = b * DF;
            const float m = m4[i];
            const float p = m * m;
            p4[i] = p;
            s_p += p;
            s_fp += p * fq;
            s_b1 += (fq >= 250.f && fq <= 650.f) ? p : 0.f;
            s_b2 += (fq >= 1000.f && fq <= 4000.f) ? p : 0.f;
            const float dm = have_prev ? (m - Mp[b]) : 0.f;
            s_fl += dm * dm;
            s_sh += p * S.tab.sharp[b];
            s_m += m;
            s_lg += logf(fmaxf(p, 1e-30f));
            if (b >= 1) {
                const float ml = M[b - 1], mr = M[b + 1];
                s_pk += fmaxf(m - 0.5f * (ml + mr), 0.f);   // prominence over the neighbours' mean (continuous)
            }
        }
        if (lane == 0) {
            const float fq = 256 * DF;
            s_fp += p256 * fq;
            s_b2 += 0.f;
            const float dm = have_prev ? (m256 - Mp[256]) : 0.f;
            s_fl += dm * dm;
            s_sh += p256 * S.tab.sharp[256];
            s_m += m256;
            s_lg += logf(fmaxf(p256, 1e-30f));
        }
        // inclusive scan of per-lane power (bins 0..255), total adds bin 256
        const float lane_p = s_p;
        const float incl = wave_scan_incl(lane_p);
        const float tot = readlane_f32(incl, 63) + p256;
        const float excl = incl - lane_p;
        const float tot_fp = wave_sum(s_fp);
        const float band1 = wave_sum(s_b1);
        const float band2 = wave_sum(s_b2);
        const float flux = sqrtf(wave_sum(s_fl) / NBINS);
        const float sharp = wave_sum(s_sh);
        const float msum = wave_sum(s_m);
        const float lgsum = wave_sum(s_lg);
        const float pksum = wave_sum(s_pk);
        const float safe = tot > 0.f ? tot : 1.0f;
        const float cen = tot_fp / safe;
        // roll-off: first bin whose inclusive cumulative power reaches p * total
        int ro[4];
        {
            const float thr[4] = {0.25f * tot, 0.50f * tot, 0.75f * tot, 0.90f * tot};
            float c = excl;
            int cand[4] = {256, 256, 256, 256};
#pragma unroll
            for (int i = 3; i >= 0; --i) { (void)i; }
            float cs[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { c += p4[i]; cs[i] = c; }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int i = 3; i >= 0; --i)
                    if (cs[i] >= thr[t]) cand[t] = 4 * lane + i;
                ro[t] = wave_min_i32(cand[t]);
            }
        }
        // second pass: central moments + entropy
        float s_e = 0.f, s_v = 0.f, s_s = 0.f, s_k = 0.f;
        const float inv = 1.0f / safe;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float fq = (4 * lane + i) * DF;
            const float pr = p4[i] * inv;
            const float d = fq - cen;
            s_e += pr > 0.f ? pr * log2f(pr) : 0.f;
            const float d2 = d * d;
            s_v += d2 * pr;
            s_s += d2 * d * pr;
            s_k += d2 * d2 * pr;
        }
        if (lane == 0) {
            const float pr = p256 * inv;
            const float d = 256 * DF - cen;
            s_e += pr > 0.f ? pr * log2f(pr) : 0.f;
            const float d2 = d * d;
            s_v += d2 * pr;
            s_s += d2 * d * pr;
            s_k += d2 * d2 * pr;
        }
        const float ent = -wave_sum(s_e);
        const float var = wave_sum(s_v);
        const float sk = wave_sum(s_s);
        const float ku = wave_sum(s_k);
        if (lane == 0) {
            const float vs = var > 0.f ? var : 1.0f;
            // sum f = DF * 256*257/2, sum f^2 = DF^2 * 256*257*513/6  (bins 0..256)
            const float sf = DF * 32896.0f;
            const float sff = DF * DF * 5625216.0f;
            S.out[0][fl] = rms;
            S.out[13][fl] = zcr;
            S.out[16][fl] = inten;
            S.out[17][fl] = powf(inten, 0.3f);
            S.out[22][fl] = band1;
            S.out[23][fl] = band2;
            S.out[24][fl] = ro[0] * DF;
            S.out[25][fl] = ro[1] * DF;
            S.out[26][fl] = ro[2] * DF;
            S.out[27][fl] = ro[3] * DF;
            S.out[28][fl] = flux;
            S.out[29][fl] = cen;
            S.out[30][fl] = ent;
            S.out[31][fl] = var;
            S.out[32][fl] = sk / (vs * sqrtf(vs));
            S.out[33][fl] = ku / (vs * vs);
            S.out[34][fl] = (NBINS * tot_fp - sf * tot) / (NBINS * sff - sf * sf);
            S.out[35][fl] = sharp / safe;
            S.out[36][fl] = pksum / (msum > 0.f ? msum : 1.0f);
            S.out[37][fl] = expf(lgsum / NBINS) / fmaxf(tot / NBINS, 1e-30f);
        }
        cur ^= 1;
        have_prev = true;
    }
    __syncthreads();

    // ---- coalesced contour-major store of the run ----
    const int64_t fbase = frame_off[clip] + f0;
    const int nvalid = (int)min((int64_t)RUN, n_fr - f0);
    const float qnan = __int_as_float(0x7fc00000);
    for (int idx = tid; idx < NLLD * RUN; idx += 256) {
        const int i = idx / RUN, t = idx % RUN;
        if (t >= nvalid) continue;
        const bool built = !(i == 14 || i == 15 || (i >= 18 && i <= 21));
        lld[(int64_t)i * total_frames + fbase + t] = built ? S.out[i][t] : qnan;
    }
}

}  // namespace smile
}  // namespace rsaf

using namespace rsaf;
using namespace rsaf::smile;

extern "C" {

int64_t rsaf_smile_n_frames(int64_t n_samples) {
    return n_samples < FRAME ? 0 : (n_samples - FRAME) / HOP + 1;
}

int rsaf_init_device(int device) {
    RSAF_CHECK_HIP(hipSetDevice(device));
    const Tables* t = nullptr;
    return get_tables(&t);
}

int rsaf_smile_lld_batch(const float* wav, const int64_t* clip_off, const int64_t* frame_off,
                         int n_clips, int64_t max_clip_frames, int64_t total_frames, float* lld,
                         rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && n_clips <= 65535, "n_clips must be in [0, 65535] per call");
    RSAF_CHECK_ARG(total_frames >= 0 && max_clip_frames >= 0, "negative frame count");
    if (n_clips == 0 || total_frames == 0 || max_clip_frames == 0) return RSAF_OK;
    RSAF_CHECK_ARG(wav && clip_off && frame_off && lld, "NULL pointer");
    const Tables* tab = nullptr;
    int rc = get_tables(&tab);
    if (rc != RSAF_OK) return rc;
    const int64_t runs = (max_clip_frames + RUN - 1) / RUN;
    RSAF_CHECK_ARG(runs <= 0x7fffffffLL, "clip too long");
    static bool attr_set[64] = {false};
    int dev = 0;
    RSAF_CHECK_HIP(hipGetDevice(&dev));
    if (!attr_set[dev]) {
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)smile_lld_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem)));
        attr_set[dev] = true;
    }
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes: every sample read once (4 B) + the LLD rows written (38 * 4 B per frame)
    ProfScope prof("smile_lld", s, 0.0, 0.0);
    dim3 grid((unsigned)runs, (unsigned)n_clips);
    hipLaunchKernelGGL(smile_lld_kernel, grid, dim3(256), sizeof(Smem), s, wav, clip_off, frame_off,
                       total_frames, lld, tab);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

}  // extern "C"
