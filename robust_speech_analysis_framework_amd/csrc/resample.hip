// Sample-rate conversion to 16 kHz in front of the extractors for gfx950.
//   rsaf_resample_sinc_hann : torchaudio.transforms.Resample defaults (sinc_interp_hann, width 6, rolloff 0.99),
//                             the resampler of src/foundation_model_extractor.py:93-94.  Polyphase FIR in fp32,
//                             one thread per output sample; the host passes each phase's non-zero taps.
//   rsaf_resample_praat     : Sound.resample(16000, 50) of src/mshds_extractor.py:419 as one raised-cosine windowed
//                             sinc (cut-off at the lower Nyquist, half-width precision + 1 input samples), fp64 math.
// Both are HBM-bound streaming kernels: 4 B read per input sample (taps and neighbours come from L1/L2), 4 B written.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

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

__global__ __launch_bounds__(256) void praat_kernel(const float* __restrict__ x, int64_t n_in, double fs_in, double fs_out,
                                                    int depth, float* __restrict__ out, int64_t n_out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const double dx_in = 1.0 / fs_in, dx_out = 1.0 / fs_out;
    const double duration = (double)n_in * dx_in;
    const double x1o = 0.5 * (duration - (double)(n_out - 1) * dx_out);
    const double pos = (x1o + (double)o * dx_out - 0.5 * dx_in) / dx_in;      // real index into the input
    const double base = floor(pos), frac = pos - base;
    const double ratio = fs_out < fs_in ? fs_out / fs_in : 1.0;               // relative cut-off
    const double d0 = frac + (double)depth;                                    // distance to tap k = -depth
    double ss, sc, ws, wc, rs, rc, vs, vc;
    sincos(PI * ratio * d0, &ss, &sc);
    sincos(PI * d0 / (depth + 1.0), &ws, &wc);
    sincos(PI * ratio, &rs, &rc);
    sincos(PI / (depth + 1.0), &vs, &vc);
    const int64_t j0 = (int64_t)base - depth;
    double acc = 0.0;
    // taps k = -depth .. depth + 1 cover every |d| <= depth + 1 for any frac in [0, 1): the window reaches zero
    // at both ends, so the result is continuous in the position (no knife edge at integer ratios)
    for (int k = 0; k <= 2 * depth + 1; ++k) {
        const double d = d0 - (double)k;
        const int64_t j = j0 + k;
        // the two taps around the position (|d| < 1) take sin() directly: the rotated value carries an absolute
        // error of ~1e-15 that the division by a tiny d would blow up
        const double sn = (k == depth || k == depth + 1) ? sin(PI * ratio * d) : ss;
        double w = d == 0.0 ? ratio : sn / (PI * d);
        w *= 0.5 + 0.5 * wc;
        if (j >= 0 && j < n_in && fabs(d) <= depth + 1.0) acc += (double)x[j] * w;
        const double s2 = ss * rc - sc * rs, c2 = sc * rc + ss * rs;           // rotate both angles one tap back
        ss = s2; sc = c2;
        const double w2 = ws * vc - wc * vs, u2 = wc * vc + ws * vs;
        ws = w2; wc = u2;
    }
    out[o] = (float)acc;
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

int rsaf_resample_praat(const float* in, int64_t n_in, double fs_in, double fs_out, int precision, float* out,
                        int64_t n_out, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_in >= 0 && n_out >= 0 && fs_in > 0.0 && fs_out > 0.0 && precision >= 1 && precision <= 4096, "bad arguments");
    if (n_out == 0) return RSAF_OK;
    RSAF_CHECK_ARG(in && out, "NULL pointer");
    RSAF_CHECK_ARG((n_out + 255) / 256 <= 0x7fffffffLL, "output too long for one launch");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("resample_praat", s, 0.0, 4.0 * (double)(n_in + n_out));
    hipLaunchKernelGGL(resample::praat_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, in, n_in, fs_in, fs_out,
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
