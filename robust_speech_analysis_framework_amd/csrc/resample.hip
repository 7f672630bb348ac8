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
#include "praat_lowpass.h"
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

// ---- whole-sound FFT low-pass: the three passes (praat_lowpass.h) over a batch, one sound per y index -----------
template <bool INV>
__global__ __launch_bounds__(256) void lp_cols_kernel(const float* __restrict__ in, c64* __restrict__ work_base,
                                                      double* __restrict__ out_base, LpBatch B, LpTables T) {
    extern __shared__ c64 lp_lds[];
    const LpSig sg = B.sigs ? B.sigs[blockIdx.y] : B.one;
    lp_cols_body<INV>(in, work_base, out_base, sg, (int)blockIdx.x, T, lp_lds);
}

__global__ __launch_bounds__(256) void lp_rows_kernel(c64* __restrict__ work_base, LpBatch B, LpTables T) {
    extern __shared__ c64 lp_lds[];
    const LpSig sg = B.sigs ? B.sigs[blockIdx.y] : B.one;
    lp_rows_body(work_base, sg, (int)blockIdx.x, B.upfactor, T, lp_lds, B.mode);
}

// ---- NUM_interpolate_sinc on the re-centred grid -------------------------------------------------------------
// One thread per output sample.  SRC = double (the low-passed sound) or float (rate going up: no filter).
template <typename SRC>
__global__ __launch_bounds__(256) void praat_interp_kernel(const SRC* __restrict__ y, int64_t n_in, double fs_in, double fs_out,
                                                           int depth, float* __restrict__ out, int64_t n_out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    // a sound read from a file: dx = 1 / fs, x1 = 0.5 / fs, domain [0, n / fs]; the new grid is centred in that domain
    const double dx_in = 1.0 / fs_in, dx_out = 1.0 / fs_out;
    const double xmax = (double)n_in / fs_in;
    const double x1o = 0.5 * (xmax - (double)(n_out - 1) / fs_out);
    const double x = (x1o + (double)o * dx_out - 0.5 / fs_in) / dx_in + 1.0;   // Praat's 1-based real index
    out[o] = (float)praat_interpolate_sinc(y, n_in, x, depth);
}

// Sound_upsample: output sample 2p = the ramp-filtered sound at p, 2p + 1 = the same spectrum half an input sample later
__global__ __launch_bounds__(256) void upsample_interleave_kernel(const double* __restrict__ even, const double* __restrict__ odd,
                                                                  int64_t n_in, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * n_in) return;
    out[i] = (float)((i & 1) ? odd[i >> 1] : even[i >> 1]);
}

}  // namespace resample
}  // namespace rsaf

namespace rsaf {
namespace resample {

// device tables of the low-pass transform of nfft samples, cached per (device, nfft); angles reduced on the host
int lp_tables(int64_t nfft, LpTables* out) {
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

int64_t rsaf_praat_lowpass_max_samples(void) { return ((int64_t)1 << resample::LP_LG_MAX) - 2 * resample::ANTI_TURN_AROUND; }

int64_t rsaf_resample_praat_work_bytes(int64_t n_in, double fs_in, double fs_out) {
    if (n_in <= 0 || !(fs_in > 0.0) || !(fs_out > 0.0)) return 0;
    const double upfactor = fs_out * (1.0 / fs_in);
    const bool doubling = fabs(upfactor - 2.0) < 1e-6;  // Sound_upsample
    if (upfactor >= 1.0 && !doubling) return 0;
    int64_t nfft = 1;
    while (nfft < n_in + 2 * resample::ANTI_TURN_AROUND) nfft *= 2;
    // nfft / 2 complex numbers + the filtered sound (fp64): low-passed, or the even and the odd output samples
    return nfft * 8 + n_in * 8 * (doubling ? 2 : 1);
}

}  // extern "C"

// the three passes of the low-pass over a batch (sigs on the device, n_sigs of them) or over `one` sound
static int launch_lowpass(const float* in, double* out, resample::c64* work, const resample::LpSig* sigs, int n_sigs,
                          const resample::LpSig& one, int lg_max, double upfactor, hipStream_t s, int mode = resample::LP_LOWPASS) {
    using namespace resample;
    RSAF_CHECK_ARG(lg_max >= 11 && lg_max <= LP_LG_MAX, "sound longer than 2^26 - 2000 samples: the low-pass transform does not fit its two LDS passes");
    LpTables T;
    const int lg_tab = lg_max + (mode == LP_UPSAMPLE_ODD ? 1 : 0);   // the half-sample turn needs e^(-2 pi i k / (2 nfft))
    {
        const int rc = lp_tables((int64_t)1 << lg_tab, &T);
        if (rc != RSAF_OK) return rc;
    }
    T.lg_max = lg_tab;
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
    B.mode = mode;
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
    if (fabs(upfactor - 2.0) < 1e-6) {                  // Sound_resample hands a doubling of the rate to Sound_upsample
        RSAF_CHECK_ARG(n_out == 2 * n_in, "Sound_upsample writes 2 n samples");
        int64_t nfft = 1;
        int lg = 0;
        while (nfft < n_in + 2 * resample::ANTI_TURN_AROUND) { nfft *= 2; ++lg; }
        RSAF_CHECK_ARG(lg <= resample::LP_LG_MAX, "sound longer than 2^26 - 2000 samples: the transform does not fit its two LDS passes");
        RSAF_CHECK_ARG(work && work_bytes >= rsaf_resample_praat_work_bytes(n_in, fs_in, fs_out), "workspace missing or too small");
        resample::c64* wk = (resample::c64*)work;
        double* even = (double*)(wk + nfft / 2);
        double* odd = even + n_in;
        ProfScope prof("resample_praat", s, 4.0 * 2.5 * (double)nfft * (double)(lg - 1), 12.0 * 8.0 * (double)nfft + 4.0 * (double)(n_in + n_out));
        resample::LpSig one{};
        one.n = (int)n_in;
        one.lg = lg;
        int rc = launch_lowpass(in, even, wk, nullptr, 1, one, lg, 0.5, s, resample::LP_UPSAMPLE_EVEN);
        if (rc != RSAF_OK) return rc;
        rc = launch_lowpass(in, odd, wk, nullptr, 1, one, lg, 0.5, s, resample::LP_UPSAMPLE_ODD);
        if (rc != RSAF_OK) return rc;
        hipLaunchKernelGGL(resample::upsample_interleave_kernel, dim3(grid), dim3(256), 0, s, (const double*)even, (const double*)odd,
                           n_in, out);
        RSAF_CHECK_HIP(hipGetLastError());
        return RSAF_OK;
    }
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
    RSAF_CHECK_ARG(lg <= resample::LP_LG_MAX, "sound longer than 2^26 - 2000 samples: the low-pass transform does not fit its two LDS passes");
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
