// Sequential tail of the openSMILE pitch chain for gfx950: cPitchSmootherViterbi -> cValbasedSelector ->
// cPitchJitter (Androids.conf:190-255 of the reference, reached through src/opensmile_extractor.py:62-87).
// The per-frame candidates come from smile_lld_kernel (cSpecScale + cPitchShs).
//
// Everything here is float64, like the frame kernel: the path costs, the lag decisions and the rows they produce then
// coincide with the float64 restatement (oracle/smile_oracle.py) except on ties of ~1e-15 (see smile_lld.hip).
//
// smile_viterbi_kernel: one wave per clip.  Lane (j, i) = (lane >> 3, lane & 7) owns the transition from state i of
//   the previous frame to state j of the current one (states 0..5 = candidate slots, 6 = unvoiced); the minimum over
//   the predecessors is three exchange steps on a 64-bit key: the bits of the (non-negative) cost with the predecessor
//   index in the three lowest mantissa bits (costs within 8 ulp of a double tie towards the lowest index, as an exact
//   tie does in the oracle).  Back pointers go to a per-frame byte record; the
//   fixed-lag decisions (bufferLength = 30) are then read off in parallel, one lane per frame, followed by the
//   energy gate.  Latency bound by construction: one dependent step per frame.
// smile_jitter_kernel: waveform matching is sequential inside a run of voiced frames and independent between runs:
//   JWAVES single-wave workgroups per clip take the runs round-robin.  Lanes = lags of the normalised cross-correlation,
//   float64 accumulation (the samples are exact in double, so lag decisions do not depend on summation order), samples
//   staged in a sliding LDS window.
//
// Semantics = oracle/smile_oracle.py (viterbi_smooth, energy_gate, jitter_shimmer), parity unpinned.
#include <cmath>

#include "rsaf_common.h"

namespace rsaf {
namespace smile {

constexpr int NCAND = RSAF_SMILE_NCAND;
constexpr int NLLD = RSAF_SMILE_NLLD;
constexpr double W_TVV = 10.0, W_TVVD = 5.0, W_TVUV = 10.0, W_THR = 4.0, W_TUU = 0.0, W_LOCAL = 2.0;
constexpr double V_CUTOFF = 0.7;
constexpr int BUFLEN = 30;
constexpr double ENERGY_GATE = 0.001;
constexpr double INF = 1e30;
constexpr int JWAVES = 8;
constexpr int VCH = 64;               // frames whose local costs are computed at once ahead of the Viterbi steps
constexpr int WCAP = 4096;            // floats of the sliding sample window
constexpr int CCMAX = 1024;           // lags kept for the parabolic refinement (0.5 * fs / 52 + 1 <= 632 up to 65 kHz)

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, m, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ __launch_bounds__(64) void smile_viterbi_kernel(const double* __restrict__ cand,
                                                           const int64_t* __restrict__ frame_off, int64_t total_frames,
                                                           double* __restrict__ lld, unsigned char* __restrict__ back) {
    __shared__ double dd[64];
    __shared__ double s_local[2][VCH * 8], s_l2f[2][VCH * 8];
    __shared__ unsigned char s_has[2][VCH * 8];
    const int clip = blockIdx.x;
    const int lane = threadIdx.x;
    const int j = lane >> 3, i = lane & 7;
    const int64_t fo = frame_off[clip];
    const int T = (int)(frame_off[clip + 1] - fo);
    if (T <= 0) return;
    const double2* c2 = reinterpret_cast<const double2*>(cand) + fo * NCAND;
    unsigned char* bk = back + fo * 8;

    // A frame's local costs (two logarithms per state) do not depend on the path: they are computed for VCH frames at a
    // time with every lane busy (8 (frame, state) pairs per lane), one chunk ahead of the dependent steps, into LDS; the
    // candidate loads of the chunk after that are in flight meanwhile.
    constexpr int PP = VCH * 8 / 64;
    double2 cv[PP];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int pr = lane + 64 * u, t = t0 + (pr >> 3), st = pr & 7;
            cv[u] = (st < NCAND && t < T) ? c2[(int64_t)t * NCAND + st] : make_double2(0.0, 0.0);
        }
    };
    auto derive_to = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int pr = lane + 64 * u, st = pr & 7;
            const bool voiced = st < NCAND;
            const bool has = voiced ? cv[u].x > 0.0 : (st == NCAND);
            double vbest = voiced && has ? cv[u].y : 0.0;     // highest voicing among the frame's candidates (8-lane group)
            vbest = fmax(vbest, __shfl_xor(vbest, 1, 64));
            vbest = fmax(vbest, __shfl_xor(vbest, 2, 64));
            vbest = fmax(vbest, __shfl_xor(vbest, 4, 64));
            double local;
            if (voiced) local = W_LOCAL * -log(fmax(cv[u].y, 1e-3)) + (cv[u].y < V_CUTOFF ? W_THR : 0.0);
            else local = W_LOCAL * -log(fmax(1.0 - vbest, 1e-3)) + (vbest >= V_CUTOFF ? W_THR : 0.0);
            s_local[buf][pr] = local;
            s_l2f[buf][pr] = (voiced && has) ? log2(cv[u].x) : 0.0;
            s_has[buf][pr] = has ? 1 : 0;
        }
    };
    fetch(0);
    derive_to(0);
    fetch(VCH);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    double pc = INF, ps = 0.0, pl = 0.0;                   // state i of the previous frame: cost, slope, log2 f0
    const int jj = j < 7 ? j : 7;
#pragma unroll 1
    for (int t0 = 0; t0 < T; t0 += VCH) {
        const int buf = (t0 / VCH) & 1;
        if (t0 + VCH < T) {
            derive_to(buf ^ 1);
            fetch(t0 + 2 * VCH);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const int tend = t0 + VCH < T ? t0 + VCH : T;
#pragma unroll 1
        for (int t = t0; t < tend; ++t) {
            const int li = (t - t0) * 8 + jj;
            const double local = s_local[buf][li], l2f = s_l2f[buf][li];
            const bool has = j <= NCAND && s_has[buf][li];
            double newcost, slope = 0.0;
            int istar = 0;
            if (t == 0) {
                newcost = has ? local : INF;
            } else {
                const bool ui = i == NCAND, uj = j == NCAND;
                const double d = (ui || uj) ? 0.0 : l2f - pl;
                const double tr = (ui && uj) ? W_TUU : ((ui || uj) ? W_TVUV : W_TVV * fabs(d) + W_TVVD * fabs(d - ps));
                const double c = (i <= NCAND && pc < INF) ? pc + tr : INF;
                unsigned long long key = ((unsigned long long)__double_as_longlong(c) & ~7ull) | (unsigned long long)i;
                unsigned long long o1 = shfl_xor_u64(key, 1); key = o1 < key ? o1 : key;
                o1 = shfl_xor_u64(key, 2); key = o1 < key ? o1 : key;
                o1 = shfl_xor_u64(key, 4); key = o1 < key ? o1 : key;
                istar = (int)(key & 7ull);
                const double cmin = __shfl(c, 8 * j + istar, 64);     // the winner's exact cost (the key only ranks)
                newcost = (has && cmin < INF) ? cmin + local : INF;
                dd[lane] = d;
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                slope = (j == NCAND || istar == NCAND) ? 0.0 : dd[8 * j + istar];
                __builtin_amdgcn_wave_barrier();
            }
            // renormalise by the best state; remember which one it is (ties: lowest state)
            double mn = INF;
            int be = 0;
#pragma unroll
            for (int st = NCAND; st >= 0; --st) {
                const double cs = readlane_f64(newcost, 8 * st);
                if (cs <= mn) { mn = cs; be = st; }
            }
            if (i == 0 && j <= NCAND) bk[(int64_t)t * 8 + j] = (unsigned char)istar;
            if (lane == 56) bk[(int64_t)t * 8 + 7] = (unsigned char)be;
            newcost = newcost < INF ? newcost - mn : INF;
            // state s's values to every lane whose predecessor index is s
            const int srcl = 8 * i;
            pc = __shfl(newcost, srcl, 64);
            ps = __shfl(slope, srcl, 64);
            pl = __shfl(l2f, srcl, 64);
        }
    }
    __threadfence();
    __builtin_amdgcn_wave_barrier();
    // fixed-lag decisions, one lane per frame, then the energy gate (cValbasedSelector on pcm_RMSenergy)
    const double* rms = lld + fo;
    double* f0row = lld + (int64_t)14 * total_frames + fo;
    double* vrow = lld + (int64_t)15 * total_frames + fo;
    for (int t = lane; t < T; t += 64) {
        const int e = min(t + BUFLEN - 1, T - 1);
        int s = bk[(int64_t)e * 8 + 7];
        for (int u = e; u > t; --u) s = bk[(int64_t)u * 8 + s];
        double F, V;
        if (s == NCAND) {
            double vb = 0.0;
#pragma unroll
            for (int k = 0; k < NCAND; ++k) {
                const double2 c = c2[(int64_t)t * NCAND + k];
                if (c.x > 0.0) vb = fmax(vb, c.y);
            }
            F = 0.0; V = vb;
        } else {
            const double2 c = c2[(int64_t)t * NCAND + s];
            F = c.x; V = c.y;
        }
        const bool keep = rms[t] >= ENERGY_GATE;
        f0row[t] = keep ? F : 0.0;
        vrow[t] = keep ? V : 0.0;
    }
}

__device__ __forceinline__ double wave_max_f64(double x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x = fmax(x, __shfl_xor(x, o, 64));
    return x;
}
__device__ __forceinline__ float wave_max_f32x(float x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}
__device__ __forceinline__ float wave_min_f32x(float x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x = fminf(x, __shfl_xor(x, o, 64));
    return x;
}

__global__ __launch_bounds__(64) void smile_jitter_kernel(const float* __restrict__ wav,
                                                          const int64_t* __restrict__ clip_off,
                                                          const int64_t* __restrict__ frame_off, int64_t total_frames,
                                                          int fs, int frame, int hop, double* __restrict__ lld) {
    __shared__ float xs[WCAP];
    __shared__ double ccb[CCMAX];
    const int clip = blockIdx.y;
    const int my = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t s0 = clip_off[clip];
    const int64_t n_samp = clip_off[clip + 1] - s0;
    const int64_t fo = frame_off[clip];
    const int T = (int)(frame_off[clip + 1] - fo);
    if (T <= 0) return;
    const float* src = wav + s0;
    const double* f0row = lld + (int64_t)14 * total_frames + fo;
    double* o_jl = lld + (int64_t)18 * total_frames + fo;
    double* o_jd = lld + (int64_t)19 * total_frames + fo;
    double* o_sh = lld + (int64_t)20 * total_frames + fo;
    double* o_hn = lld + (int64_t)21 * total_frames + fo;

    int run_id = 0;
    bool carry = false;                                    // frame base-1 voiced
    int64_t wbase = -1;                                    // first sample held in xs (-1: nothing staged)
#pragma unroll 1
    for (int base = 0; base < T; base += 64) {
        const int tl = base + lane;
        const bool voiced = tl < T && f0row[tl] > 0.0;
        if (my == 0 && tl < T && !voiced) { o_jl[tl] = 0.0; o_jd[tl] = 0.0; o_sh[tl] = 0.0; o_hn[tl] = 0.0; }
        const unsigned long long vm = __ballot(voiced);
        unsigned long long starts = vm & ~((vm << 1) | (carry ? 1ull : 0ull));
        carry = (vm >> 63) & 1ull;
#pragma unroll 1
        while (starts) {
            const int b = __ffsll((long long)starts) - 1;
            starts &= starts - 1;
            const int mine = (run_id++ % JWAVES) == my;
            if (!mine) continue;
            // ---- one run of voiced frames starting at frame base + b ----
            double pos = (double)(base + b) * hop;
            bool hasT = false, hasD = false, dead = false;
            double prevT = 0, prevD = 0, prevA = 0;
            double l_jl = 0.0, l_jd = 0.0, l_sh = 0.0, l_hn = 0.0;
#pragma unroll 1
            for (int t = base + b; t < T; ++t) {
                const double f0f = f0row[t];
                if (!(f0f > 0.0)) break;
                const double T0 = (double)fs / f0f;
                const int lo = (int)ceil(0.75 * T0), hi = (int)floor(1.25 * T0);
                const int W = (int)floor(T0 + 0.5);
                const double end = (double)(t + 1) * hop;
                int nT = 0, ndT = 0, ndD = 0;
                double sT = 0, sA = 0, sC = 0, sdT = 0, sdD = 0, sdA = 0;
#pragma unroll 1
                while (!dead && pos < end) {
                    const int64_t p = (int64_t)floor(pos);
                    if (p + W + hi > n_samp || lo < 1 || hi < lo || hi - lo + 1 > CCMAX || W + hi > WCAP) { dead = true; break; }
                    if (wbase < 0 || p < wbase || p + W + hi > wbase + WCAP) {
                        __builtin_amdgcn_wave_barrier();
                        wbase = p;
                        for (int k = lane; k < WCAP; k += 64) {
                            const int64_t si = wbase + k;
                            xs[k] = si < n_samp ? src[si] : 0.f;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    const float* x = xs + (p - wbase);
                    double e0 = 0;
                    for (int n = lane; n < W; n += 64) { const double a = x[n]; e0 += a * a; }
                    e0 = wave_sum_f64(e0);
                    const int NL = hi - lo + 1;
                    double best = -2.0;
#pragma unroll 1
                    for (int r0 = 0; r0 < NL; r0 += 64) {
                        const int q = r0 + lane;
                        const int tau = lo + min(q, NL - 1);
                        double num = 0, e1 = 0;
                        const float* y = x + tau;
#pragma unroll 4
                        for (int n = 0; n < W; ++n) {
                            const double a = x[n], bb = y[n];
                            num += a * bb;
                            e1 += bb * bb;
                        }
                        const double cc = (e0 > 0 && e1 > 0) ? num / sqrt(e0 * e1) : 0.0;
                        if (q < NL) { ccb[q] = cc; best = fmax(best, cc); }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    best = wave_max_f64(best);
                    // first lag that attains the maximum
                    int qbest = 0x7fffffff;
                    for (int q = lane; q < NL; q += 64)
                        if (ccb[q] == best) { qbest = q; break; }
                    qbest = wave_min_i32(qbest);
                    const int tau = lo + qbest;
                    double Tp = (double)tau, cs = best;
                    if (qbest > 0 && qbest < NL - 1) {
                        const double y1 = ccb[qbest - 1], y2 = best, y3 = ccb[qbest + 1];
                        const double den = y1 - 2.0 * y2 + y3;
                        if (den < 0.0) {
                            Tp = tau + 0.5 * (y1 - y3) / den;
                            cs = y2 - 0.125 * (y1 - y3) * (y1 - y3) / den;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    float amx = -INFINITY, amn = INFINITY;
                    for (int n = lane; n < tau; n += 64) { amx = fmaxf(amx, x[n]); amn = fminf(amn, x[n]); }
                    const double A = (double)wave_max_f32x(amx) - (double)wave_min_f32x(amn);
                    if (hasT) {
                        const double d = Tp - prevT;
                        sdT += fabs(d); ++ndT;
                        if (hasD) { sdD += fabs(d - prevD); ++ndD; }
                        prevD = d; hasD = true;
                        sdA += fabs(A - prevA);
                    }
                    prevT = Tp; prevA = A; hasT = true;
                    sT += Tp; sA += A; sC += cs; ++nT;
                    pos += Tp;
                }
                if (nT > 0) {
                    const double mT = sT / nT, mA = sA / nT;
                    double c = sC / nT;
                    c = fmin(fmax(c, 1e-3), 1.0 - 1e-6);
                    l_jl = ndT > 0 ? (sdT / ndT) / mT : 0.0;
                    l_jd = ndD > 0 ? (sdD / ndD) / mT : 0.0;
                    l_sh = (ndT > 0 && mA > 0.0) ? (sdA / ndT) / mA : 0.0;
                    l_hn = log(c / (1.0 - c));
                }
                if (lane == 0) { o_jl[t] = l_jl; o_jd[t] = l_jd; o_sh[t] = l_sh; o_hn[t] = l_hn; }
            }
        }
    }
}

}  // namespace smile
}  // namespace rsaf

using namespace rsaf;
using namespace rsaf::smile;

extern "C" {

int64_t rsaf_smile_workspace_bytes(int64_t total_frames) {
    if (total_frames < 0) return -1;
    // candidates (f0, voicing) x 6 per frame (float64) + the 8-byte back-pointer record per frame
    return total_frames * (int64_t)(NCAND * 2 * sizeof(double) + 8) + 256;
}

int rsaf_smile_pitch_track(const float* wav, const int64_t* clip_off, const int64_t* frame_off, int n_clips,
                           int64_t total_frames, int sample_rate, const double* cand, void* back_workspace, double* lld,
                           rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n_clips >= 0 && n_clips <= 65535, "n_clips must be in [0, 65535] per call");
    if (n_clips == 0 || total_frames == 0) return RSAF_OK;
    RSAF_CHECK_ARG(wav && clip_off && frame_off && cand && back_workspace && lld, "NULL pointer");
    int frame = 0, hop = 0, nfft = 0;
    int rc = rsaf_smile_geometry(sample_rate, &frame, &hop, &nfft);
    if (rc != RSAF_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope prof("smile_viterbi", s, 0.0, 0.0);
        hipLaunchKernelGGL(smile_viterbi_kernel, dim3((unsigned)n_clips), dim3(64), 0, s, cand, frame_off, total_frames,
                           lld, static_cast<unsigned char*>(back_workspace));
        RSAF_CHECK_HIP(hipGetLastError());
    }
    {
        ProfScope prof("smile_jitter", s, 0.0, 0.0);
        hipLaunchKernelGGL(smile_jitter_kernel, dim3(JWAVES, (unsigned)n_clips), dim3(64), 0, s, wav, clip_off, frame_off,
                           total_frames, sample_rate, frame, hop, lld);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    return RSAF_OK;
}

}  // extern "C"
