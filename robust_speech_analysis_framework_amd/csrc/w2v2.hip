// Wav2Vec2 frame-embedding forward for gfx950 (fp32 end to end).
//
// Replaces, for a batch of equal-length chunks, what the reference runs per chunk at batch 1
// (src/foundation_model_extractor.py:113-116): HF Wav2Vec2FeatureExtractor normalisation
// (feature_extraction_wav2vec2.py:95) + transformers' Wav2Vec2Model.forward in eval mode
// (modeling_wav2vec2.py:254-323 conv stack, :422-434 projection, :326-379 positional conv,
// :575-608 post-LN encoder layers).  Chunking (80 000-sample windows every 64 000, per-chunk
// normalisation, duplicated overlap) stays on the host side exactly as the reference does it.
//
// Every dense contraction runs on the fp32-accurate fp16-split GEMM (gemm_f16x3.hip: operands as two fp16 planes with
// power-of-two scales, three MFMA products per term):
//   conv1..6  : channels-last activations make a k-tap/stride-2 conv a GEMM with lda = 2*C, K = k*C
//   pos-conv  : activations regrouped to [chunk][group][T+K-1 (zero padded)][C/G], one batched GEMM
//   attention : S = QK^T/sqrt(d) and O = PV as (chunk, head)-batched GEMMs on the packed qkv buffer
// conv0 (Cin = 1) + GroupNorm + GELU is recomputed in two passes (stats, apply) instead of
// materialising the un-normalised 15 999 x 512 activation; LayerNorm / softmax are one wave per row.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "gemm_f32.h"
#include "gemm_f16x3.h"

namespace rsaf {
namespace w2v2 {

struct Cfg {
    int C, Hd, L, NH, I, PK, PG;
    float eps;
};

static inline int64_t pad4(int64_t n) { return (n + 3) & ~int64_t(3); }

static const int KERN[7] = {10, 3, 3, 3, 3, 2, 2};
static const int STRD[7] = {5, 2, 2, 2, 2, 2, 2};

struct LayerOff {
    int64_t wqkv, bqkv, wo, bo, ln1g, ln1b, w1, b1, w2, b2, ln2g, ln2b;
};
struct Layout {
    int64_t conv0, gng, gnb, conv[6], fplg, fplb, fpw, fpb, posw, posb, elng, elnb;
    std::vector<LayerOff> layers;
    int64_t total;
};

static Layout make_layout(const Cfg& c) {
    Layout L;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t s = o; o += pad4(n); return s; };
    L.conv0 = take((int64_t)c.C * 10); L.gng = take(c.C); L.gnb = take(c.C);
    for (int i = 0; i < 6; ++i) L.conv[i] = take((int64_t)c.C * KERN[i + 1] * c.C);
    L.fplg = take(c.C); L.fplb = take(c.C); L.fpw = take((int64_t)c.Hd * c.C); L.fpb = take(c.Hd);
    const int cg = c.Hd / c.PG;
    L.posw = take((int64_t)c.PG * cg * c.PK * cg); L.posb = take(c.Hd);
    L.elng = take(c.Hd); L.elnb = take(c.Hd);
    for (int l = 0; l < c.L; ++l) {
        LayerOff lo;
        lo.wqkv = take((int64_t)3 * c.Hd * c.Hd); lo.bqkv = take(3 * c.Hd);
        lo.wo = take((int64_t)c.Hd * c.Hd); lo.bo = take(c.Hd);
        lo.ln1g = take(c.Hd); lo.ln1b = take(c.Hd);
        lo.w1 = take((int64_t)c.I * c.Hd); lo.b1 = take(c.I);
        lo.w2 = take((int64_t)c.Hd * c.I); lo.b2 = take(c.Hd);
        lo.ln2g = take(c.Hd); lo.ln2b = take(c.Hd);
        L.layers.push_back(lo);
    }
    L.total = o;
    return L;
}

static int check_cfg(const Cfg& c) {
    RSAF_CHECK_ARG(c.C >= 32 && c.C <= 1024 && c.C % 32 == 0 && (c.C <= 256 || c.C % 256 == 0),
                   "conv_dim must be a multiple of 32 (<= 256) or of 256 (<= 1024)");
    RSAF_CHECK_ARG(c.Hd >= 16 && c.Hd <= 1024 && c.Hd % 16 == 0, "hidden_size must be a multiple of 16, <= 1024");
    RSAF_CHECK_ARG(c.I >= 16 && c.I % 16 == 0, "intermediate_size must be a multiple of 16");
    RSAF_CHECK_ARG(c.L >= 1 && c.L <= 64, "num_hidden_layers out of range");
    RSAF_CHECK_ARG(c.NH >= 1 && c.Hd % c.NH == 0 && (c.Hd / c.NH) % 4 == 0, "head_dim must be a multiple of 4");
    RSAF_CHECK_ARG(c.PG >= 1 && c.Hd % c.PG == 0 && (c.Hd / c.PG) % 4 == 0 && c.PK >= 2 && c.PK % 2 == 0,
                   "positional conv: channels/group multiple of 4, even kernel");
    return RSAF_OK;
}

static void chunk_lengths(int len, int T[7]) {
    int n = len;
    for (int i = 0; i < 7; ++i) {
        n = n >= KERN[i] ? (n - KERN[i]) / STRD[i] + 1 : 0;
        T[i] = n;
    }
}

// The dense GEMMs run on rsaf's fp32-accurate f16x3 kernel (gemm_f16x3.hip): their operands live as two fp16 planes,
// scaled per row (encoder activations, weights) or per window (feature-encoder activations) by a power of two.
// Offsets are in floats; a planes buffer of N elements takes 2 N uint16 = N floats.
struct Workspace {
    int64_t xn, part, ab, P, Q, c6, lnfp, x, xp, y, att, attp, xg, qkv, S, ffnp, wp, total;
    int64_t wp_conv[6], wp_fp, wp_pos;
    std::vector<int64_t> wp_qkv, wp_o, wp_1, wp_2;
    // scales of the weight rows, and per weight matrix two words {max row norm, max |w|} (bit patterns, atomicMax)
    int64_t ws_conv[6], ws_fp, ws_pos, wstat;            // wstat: [6 + 2 + 4 L][2] (+ [L][2] for the ffn1 biases)
    std::vector<int64_t> ws_qkv, ws_o, ws_1, ws_2;
    // scales of the activations: per window of the current conv group (conv_scale[7][G], conv_amax[7][G]), per window of
    // the call (pos_scale, fp_amax, win_norm; wlen: the length table of an equal-window call) and per frame (ln / ffn / qkv scales)
    int64_t conv_scale, conv_amax, pos_scale, fp_amax, wlen, win_norm, s_lnfp, s_x, s_ffn, s_att, s_qkv;
    int64_t t_Tw, t_row0, t_ztab, t_rowwin;              // window tables (int32 / int64 views of the float workspace)
    int slabs, Tp, G;
};
constexpr int STAT_SLAB = 512;
constexpr int CONV_GROUP = 512;      // windows per pass of the feature encoder (its ping-pong buffers are the big ones)

static inline int64_t planes_floats(int64_t n) { return pad4(n); }

// index of a weight matrix in the wstat table
static inline int wstat_conv(int i) { return i; }                       // i = 0..5 (conv1..6)
constexpr int WSTAT_FP = 6, WSTAT_POS = 7, WSTAT_LAYER0 = 8, WSTAT_PER_LAYER = 6;   // layer l: qkv, o, ffn1, ffn2, ffn1 bias, qkv bias

// Window geometry of one call (host side): lengths are NON-INCREASING (the caller sorts), so windows of equal frame count
// are contiguous and a group of CONV_GROUP consecutive windows wastes few tile rows.
struct Rag {
    int n = 0, maxlen = 0, Tmax[7] = {};
    int64_t rows = 0;                                    // frames of all windows (rows of the encoder)
    std::vector<int> len;
    std::vector<int64_t> row0;                           // [n + 1] first encoder row of each window
    std::vector<std::pair<int, int>> tgroups;            // [begin, end) of windows with equal T[6]
    std::vector<int> T6;
};

static int make_rag(const int* len_host, int n, Rag& R) {
    R.n = n;
    R.len.assign(len_host, len_host + n);
    R.row0.assign(n + 1, 0);
    R.T6.resize(n);
    for (int w = 0; w < n; ++w) {
        RSAF_CHECK_ARG(w == 0 || R.len[w] <= R.len[w - 1], "window lengths must be non-increasing");
        int T[7];
        chunk_lengths(R.len[w], T);
        RSAF_CHECK_ARG(T[6] >= 1, "chunk shorter than the receptive field of the feature encoder");
        R.T6[w] = T[6];
        R.row0[w + 1] = R.row0[w] + T[6];
        if (w == 0) { R.maxlen = R.len[0]; for (int i = 0; i < 7; ++i) R.Tmax[i] = T[i]; }
        if (w == 0 || T[6] != R.T6[w - 1]) R.tgroups.emplace_back(w, w + 1);
        else R.tgroups.back().second = w + 1;
    }
    R.rows = R.row0[n];
    return RSAF_OK;
}

static Workspace make_ws(const Cfg& c, const Rag& R) {
    const int n = R.n;
    const int* T = R.Tmax;
    Workspace w{};
    int64_t o = 0;
    auto take = [&](int64_t k) { int64_t s = o; o += pad4(k); return s; };
    const int Tt = T[6];
    const int64_t rows = R.rows;
    const int G = n < CONV_GROUP ? n : CONV_GROUP;
    w.G = G;
    w.slabs = (T[0] + STAT_SLAB - 1) / STAT_SLAB;
    w.Tp = (int)pad4(Tt);
    w.xn = take((int64_t)G * R.maxlen);
    w.part = take((int64_t)G * w.slabs * 3 * c.C);
    w.ab = take((int64_t)G * 2 * c.C);
    w.P = take(planes_floats((int64_t)G * T[0] * c.C));
    w.Q = take(planes_floats((int64_t)G * T[1] * c.C));
    w.c6 = take(rows * c.C);
    w.lnfp = take(planes_floats(rows * c.C));
    w.x = take(rows * c.Hd);
    w.xp = take(planes_floats(rows * c.Hd));
    w.y = take(rows * c.Hd);
    w.att = take(rows * c.Hd);
    w.attp = take(planes_floats(rows * c.Hd));
    w.xg = take((int64_t)n * (Tt + c.PK - 1) * c.Hd);                     // fp32 or two fp16 planes (same size)
    w.qkv = take(rows * 3 * c.Hd);
    w.S = take((c.Hd / c.NH == 64 && Tt <= 256) ? 4 : (int64_t)n * c.NH * Tt * w.Tp);   // scores: only the three-launch attention
    w.ffnp = take(planes_floats(rows * c.I));
    // weight planes and row scales (built once per forward call)
    for (int i = 0; i < 6; ++i) {
        w.wp_conv[i] = take(planes_floats((int64_t)c.C * KERN[i + 1] * c.C));
        w.ws_conv[i] = take(c.C);
    }
    w.wp_fp = take(planes_floats((int64_t)c.Hd * c.C)); w.ws_fp = take(c.Hd);
    w.wp_pos = take(planes_floats((int64_t)c.Hd * c.PK * (c.Hd / c.PG))); w.ws_pos = take(c.Hd);
    for (int l = 0; l < c.L; ++l) {
        w.wp_qkv.push_back(take(planes_floats((int64_t)3 * c.Hd * c.Hd))); w.ws_qkv.push_back(take(3 * c.Hd));
        w.wp_o.push_back(take(planes_floats((int64_t)c.Hd * c.Hd))); w.ws_o.push_back(take(c.Hd));
        w.wp_1.push_back(take(planes_floats((int64_t)c.I * c.Hd))); w.ws_1.push_back(take(c.I));
        w.wp_2.push_back(take(planes_floats((int64_t)c.Hd * c.I))); w.ws_2.push_back(take(c.Hd));
    }
    w.wstat = take((int64_t)(WSTAT_LAYER0 + WSTAT_PER_LAYER * c.L) * 2);
    w.conv_scale = take((int64_t)7 * G);
    w.conv_amax = take((int64_t)7 * G);
    w.pos_scale = take(n); w.fp_amax = take(n); w.wlen = take(n); w.win_norm = take(n);
    w.s_lnfp = take(rows); w.s_x = take(rows); w.s_ffn = take(rows); w.s_att = take(rows); w.s_qkv = take(rows);
    // window tables (device): Tw[7][n] frames per layer, row0[n + 1] (int64), ztab[7][n][2] (int64: the GEMM's per-batch
    // {rows, output offset} of conv1..6 and of the positional conv), rowwin[rows] window of every encoder row
    w.t_Tw = take((int64_t)7 * n); w.t_row0 = take(2 * ((int64_t)n + 1)); w.t_ztab = take((int64_t)7 * n * 2 * 2);
    w.t_rowwin = take(rows);
    w.total = o;
    return w;
}

// Tw[i][w], row0, the packed-output offsets and the row -> window map from the device copy of the lengths
__global__ __launch_bounds__(256) void w2v2_tables_kernel(const int* __restrict__ len, int n, int C, int Hd, int* __restrict__ Tw,
                                                          int64_t* __restrict__ row0, int64_t* __restrict__ ztab) {
    const int KERN_[7] = {10, 3, 3, 3, 3, 2, 2}, STRD_[7] = {5, 2, 2, 2, 2, 2, 2};
    for (int w = threadIdx.x; w < n; w += 256) {
        int m = len[w];
        for (int i = 0; i < 7; ++i) { m = m >= KERN_[i] ? (m - KERN_[i]) / STRD_[i] + 1 : 0; Tw[(int64_t)i * n + w] = m; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t r = 0;
        for (int w = 0; w < n; ++w) { row0[w] = r; r += Tw[(int64_t)6 * n + w]; }
        row0[n] = r;
    }
    __syncthreads();
    // ztab[j][w] = {rows of window w, element offset of its fp32 output or -1}: j = 0..4 conv1..5 (plane outputs at the batch
    // stride), j = 5 conv6 (packed rows of C floats), j = 6 positional conv (packed rows of Hd floats)
    for (int w = threadIdx.x; w < n; w += 256) {
        for (int j = 0; j < 5; ++j) { ztab[((int64_t)j * n + w) * 2] = Tw[(int64_t)(j + 1) * n + w]; ztab[((int64_t)j * n + w) * 2 + 1] = -1; }
        ztab[((int64_t)5 * n + w) * 2] = Tw[(int64_t)6 * n + w]; ztab[((int64_t)5 * n + w) * 2 + 1] = row0[w] * C;
        ztab[((int64_t)6 * n + w) * 2] = Tw[(int64_t)6 * n + w]; ztab[((int64_t)6 * n + w) * 2 + 1] = row0[w] * Hd;
    }
}
__global__ __launch_bounds__(256) void w2v2_rowwin_kernel(const int64_t* __restrict__ row0, int* __restrict__ rowwin) {
    const int w = blockIdx.x;
    const int64_t a = row0[w], b = row0[w + 1];
    for (int64_t r = a + threadIdx.x; r < b; r += 256) rowwin[r] = w;
}
__global__ __launch_bounds__(256) void fill_i32_kernel(int* __restrict__ p, int n, int v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- per-chunk zero-mean / unit-variance normalisation -------------------------------------------
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ wav,
                                                        const int64_t* __restrict__ starts, const int* __restrict__ wlen,
                                                        int maxlen, float* __restrict__ xn) {
    __shared__ double red[4];
    __shared__ double bc;
    const float* x = wav + starts[blockIdx.x];
    const int len = wlen[blockIdx.x];                        // every window is normalised over its own samples
    float* o = xn + (int64_t)blockIdx.x * maxlen;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double s = 0.0;
    for (int i = threadIdx.x; i < len; i += 256) s += (double)x[i];
    s = wave_sum_f64(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) bc = (red[0] + red[1] + red[2] + red[3]) / len;
    __syncthreads();
    const double mean = bc;
    double v = 0.0;
    for (int i = threadIdx.x; i < len; i += 256) { const double d = (double)x[i] - mean; v += d * d; }
    v = wave_sum_f64(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    const float var = (float)((red[0] + red[1] + red[2] + red[3]) / len);
    const float mu = (float)mean;
    const float sd = sqrtf(var + 1e-7f);
    for (int i = threadIdx.x; i < len; i += 256) o[i] = (x[i] - mu) / sd;
}

// ---- conv0 (1 -> C, k = 10, s = 5) + GroupNorm(C groups) + GELU, two passes ----------------------
__device__ __forceinline__ unsigned short f16_bits_w(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }
// xs = hi + lo (+ 2^-22 |xs|): the two fp16 planes of a value that already carries its power-of-two scale (gemm_f16x3.hip)
__device__ __forceinline__ void split2_w(float xs, unsigned short& h, unsigned short& l) {
    const _Float16 hh = (_Float16)xs;
    h = f16_bits_w(hh);
    l = f16_bits_w((_Float16)(xs - (float)hh));
}

// thread <-> channel(s); the 10 samples of a frame are wave-uniform (scalar loads).  Statistics pass: per slab and channel
// the sum, the sum of squares and the largest |y| (the bound behind the window's plane scale).  Apply pass: the output
// goes out as the two fp16 planes of GELU(a y + b) * scale[window] (A operand of conv1).
template <int CPT, bool APPLY>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ xn, const float* __restrict__ w0,
                                                    float* __restrict__ part, const float* __restrict__ ab,
                                                    const float* __restrict__ scale,
                                                    unsigned short* __restrict__ outp, int64_t plane, int len,
                                                    const int* __restrict__ T0w, int T0, int C, int slab, int slabs) {
    // len / T0: the longest window's samples / frames (strides of xn and of the output); T0w: frames of every window
    const int chunk = blockIdx.y, sl = blockIdx.x;
    const int t0 = sl * slab, t1 = min(T0w[chunk], t0 + slab);
    const float* __restrict__ x = xn + (int64_t)chunk * len;
    float wr[CPT][10], a[CPT], b[CPT], s[CPT], q[CPT], mx[CPT];
    int ch[CPT];
    const float sc = APPLY ? scale[chunk] : 1.0f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        ch[k] = CPT * threadIdx.x + k;                      // adjacent channels: the planes go out as packed pairs
#pragma unroll
        for (int j = 0; j < 10; ++j) wr[k][j] = w0[ch[k] * 10 + j];
        s[k] = 0.f; q[k] = 0.f; mx[k] = 0.f;
        if (APPLY) { a[k] = ab[((int64_t)chunk * 2 + 0) * C + ch[k]]; b[k] = ab[((int64_t)chunk * 2 + 1) * C + ch[k]]; }
    }
    for (int t = t0; t < t1; ++t) {
        float xv[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) xv[j] = x[5 * t + j];
        unsigned short hh[CPT], ll[CPT];
        float gv[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < 10; ++j) y = fmaf(wr[k][j], xv[j], y);
            if (APPLY) {
                gv[k] = fmaf(y, a[k], b[k]);
            } else {
                s[k] += y; q[k] += y * y; mx[k] = fmaxf(mx[k], fabsf(y));
            }
        }
        if (APPLY) {                                        // GELU two channels per packed instruction (gemm_f16x3.h)
#pragma unroll
            for (int k = 0; k < CPT; k += 2) {
                if (k + 1 < CPT) {
                    const gelu_f32x2 g2 = gelu_pair(gelu_f32x2{gv[k], gv[k + 1]});
                    split2_w(g2.x * sc, hh[k], ll[k]);
                    split2_w(g2.y * sc, hh[k + 1], ll[k + 1]);
                } else {
                    const gelu_f32x2 g2 = gelu_pair(gelu_f32x2{gv[k], gv[k]});
                    split2_w(g2.x * sc, hh[k], ll[k]);
                }
            }
        }
        if (APPLY) {
            const int64_t o = ((int64_t)chunk * T0 + t) * C + ch[0];
            if (CPT % 2 == 0) {                             // 4-byte stores of channel pairs (C and ch[0] are even)
#pragma unroll
                for (int k = 0; k < CPT; k += 2) {
                    *reinterpret_cast<unsigned*>(outp + o + k) = hh[k] | ((unsigned)hh[k + 1] << 16);
                    *reinterpret_cast<unsigned*>(outp + plane + o + k) = ll[k] | ((unsigned)ll[k + 1] << 16);
                }
            } else {
#pragma unroll
                for (int k = 0; k < CPT; ++k) { outp[o + k] = hh[k]; outp[plane + o + k] = ll[k]; }
            }
        }
    }
    if (!APPLY) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            part[(((int64_t)chunk * slabs + sl) * 3 + 0) * C + ch[k]] = s[k];
            part[(((int64_t)chunk * slabs + sl) * 3 + 1) * C + ch[k]] = q[k];
            part[(((int64_t)chunk * slabs + sl) * 3 + 2) * C + ch[k]] = mx[k];
        }
    }
}

// GroupNorm coefficients per (window, channel) and the window's bound: |GELU(a y + b)| <= |a| max|y| + |b|
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ g,
                                                          const float* __restrict__ be, float* __restrict__ ab,
                                                          unsigned* __restrict__ amax, int n, int C, int slabs,
                                                          const int* __restrict__ T0w) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * C) return;
    const int chunk = (int)(i / C), c = (int)(i % C);
    const int T0 = T0w[chunk];
    double s = 0.0, q = 0.0;
    float mx = 0.f;
    for (int sl = 0; sl < slabs; ++sl) {
        s += (double)part[(((int64_t)chunk * slabs + sl) * 3 + 0) * C + c];
        q += (double)part[(((int64_t)chunk * slabs + sl) * 3 + 1) * C + c];
        mx = fmaxf(mx, part[(((int64_t)chunk * slabs + sl) * 3 + 2) * C + c]);
    }
    const double mean = s / T0;
    double var = q / T0 - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + 1e-5);
    const double a = (double)g[c] * rstd;
    const float af = (float)a, bf = (float)((double)be[c] - mean * a);
    ab[((int64_t)chunk * 2 + 0) * C + c] = af;
    ab[((int64_t)chunk * 2 + 1) * C + c] = bf;
    atomicMax(amax + chunk, __float_as_uint((fabsf(af) * mx + fabsf(bf)) * 1.000001f));
}

// ---- LayerNorm over the last dim (optionally of x + r), one wave per row, D <= 1024 -----------------
// Optional outputs for the GEMM that reads the result: the row as two fp16 planes times the power of two that puts the
// row's largest magnitude into [2^14, 2^15) (scale_out[row]: exact, the wave holds the whole row), and the scale the NEXT
// GEMM's plane output may use for this row (bound_scale_out): |GELU(y W^T + b)| <= |y|_2 max_n |w_n|_2 + max |b|.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                        const float* __restrict__ g, const float* __restrict__ b,
                                                        float* __restrict__ out, int64_t rows, int D, float eps,
                                                        const int64_t* __restrict__ out_row_start, const int* __restrict__ rowwin,
                                                        const int64_t* __restrict__ row0,
                                                        unsigned short* __restrict__ planes, int64_t plane, int panel,
                                                        float* __restrict__ scale_out, const unsigned* __restrict__ bound_w,
                                                        const unsigned* __restrict__ bound_b, float* __restrict__ bound_scale_out,
                                                        unsigned* __restrict__ win_norm) {
    // panel != 0: the planes go out in the k16-panel layout of `rows` rows (gemm_f16x3.h), staged through LDS so that the
    // four rows of the workgroup leave as full 128-byte lines per panel (scattering 8-byte pieces from the row layout cost
    // this kernel + 77 %)
    __shared__ __attribute__((aligned(16))) unsigned short stg[2][4][1024];
    const int64_t row_raw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool valid = row_raw < rows;
    if (!valid && !(planes && panel)) return;
    const int64_t row = valid ? row_raw : rows - 1;
    const int lane = threadIdx.x & 63;
    // optional scatter: frame t of window w lands at output row out_row_start[w] + t (vstack order)
    int64_t orow = row;
    if (out_row_start) { const int w = rowwin[row]; orow = out_row_start[w] + (row - row0[w]); }
    const int D4 = D >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x + row * D);
    const float4* r4 = r ? reinterpret_cast<const float4*>(r + row * D) : nullptr;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < D4) {
            v[i] = x4[idx];
            if (r4) { const float4 t = r4[idx]; v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w; }
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        if (idx < D4) {
            const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / D + eps);
    float4* o4 = out ? reinterpret_cast<float4*>(out + orow * D) : nullptr;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float mx = 0.f, n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        if (idx < D4) {
            const float4 gg = g4[idx], bb = b4[idx];
            const float4 y = make_float4((v[i].x - mean) * rstd * gg.x + bb.x, (v[i].y - mean) * rstd * gg.y + bb.y,
                                         (v[i].z - mean) * rstd * gg.z + bb.z, (v[i].w - mean) * rstd * gg.w + bb.w);
            if (o4 && valid) o4[idx] = y;
            v[i] = y;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w))));
            n2 += (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
        }
    }
    if (!planes) return;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float sc = f16x2_scale_for_bound(mx);
    if (valid && lane == 0) scale_out[row] = sc;
    if (bound_scale_out || win_norm) {
        const float nrm = sqrtf(wave_sum(n2)) * 1.000001f;
        if (bound_scale_out) {
            const float bound = nrm * __uint_as_float(bound_w[0]) * 1.000001f + (bound_b ? __uint_as_float(bound_b[1]) : 0.0f);
            if (valid && lane == 0) bound_scale_out[row] = f16x2_scale_for_bound(bound);
        }
        // the largest row norm of the window (behind the per-window scale of the next q / k / v projection's plane output)
        if (win_norm && valid && lane == 0) atomicMax(win_norm + rowwin[row], __float_as_uint(nrm));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        if (idx < D4) {                                    // the same values as two fp16 planes (A operand of the next GEMM)
            const float yy[4] = {v[i].x * sc, v[i].y * sc, v[i].z * sc, v[i].w * sc};
            unsigned short hh[4], ll[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) split2_w(yy[k], hh[k], ll[k]);
            const uint2 ph = make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
            const uint2 pl = make_uint2(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16));
            if (panel) {
                const int wq = threadIdx.x >> 6;
                *reinterpret_cast<uint2*>(&stg[0][wq][4 * idx]) = ph;
                *reinterpret_cast<uint2*>(&stg[1][wq][4 * idx]) = pl;
            } else if (valid) {
                unsigned short* pp = planes + row * D + 4 * idx;
                *reinterpret_cast<uint2*>(pp) = ph;
                *reinterpret_cast<uint2*>(pp + plane) = pl;
            }
        }
    }
    if (panel) {
        __syncthreads();
        const int64_t row0 = (int64_t)blockIdx.x * 4;
        const int chunks = D >> 1;                          // 16-byte pieces per plane: D / 16 panels x 4 rows x 2 halves
        for (int c = threadIdx.x; c < chunks; c += 256) {
            const int pn = c >> 3, rr = (c & 7) >> 1, hf = c & 1;
            if (row0 + rr < rows) {
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2)
                    *reinterpret_cast<uint4*>(planes + p2 * plane + (int64_t)pn * (rows * 16) + (row0 + rr) * 16 + 8 * hf) =
                        *reinterpret_cast<const uint4*>(&stg[p2][rr][16 * pn + 8 * hf]);
            }
        }
    }
}

using af32x16 = __attribute__((ext_vector_type(16))) float;

typedef __attribute__((address_space(3))) void* attn_lds_ptr;
typedef const __attribute__((address_space(1))) void* attn_glb_ptr;

// ---- fused attention on the fp16 matrix pipe (two-way fp16 splits, three products: gemm_f16x3.hip's arithmetic) --------
// For windows of at most 256 frames (every Wav2Vec2 window: T <= 249).  One workgroup = 128 queries of one (window, head);
// wave = 32 queries.  S^T = K Q^T: keys are the MFMA rows, queries the columns, so a lane holds ONE query's scores for 16 keys
// per tile: the softmax over keys is in-lane plus one exchange with lane ^ 32, and no score or probability ever goes to memory;
// all 256 scores of a query stay in registers, so the softmax is the plain two-pass form, not an online rescaling.  K and V are
// staged in blocks of 128 keys by LDS-DMA (two 32 KB buffers, the next block in flight during the multiply).
// (Rounds 1-3 ran this on the fp32 matrix pipe, v_mfma_f32_32x32x2_f32: 78 TFLOP/s; now 161.)  Every product runs on
// v_mfma_f32_32x32x16_f16:
//   * q, k, v arrive as the fp16 plane pair the q/k/v projection's epilogue wrote, scaled by ONE power of two per window
//     (s_w: the bound |x|_2 max|w_n|_2 + max|b| over the window's rows), so S = acc / s_w^2 and the value scale factors out
//     of the sum over keys; no conversion work in this kernel except for the probabilities;
//   * K blocks of 128 keys x 64 halfs x 2 planes (32 KB) by LDS-DMA, 16-byte chunks XOR-swizzled on the source address
//     (chunk ^ ((row >> 1) & 7): the ds_read_b128 fragment reads of 16 consecutive rows cover all 64 banks);
//   * the score tile's register layout is the A operand of P V (element j of lane half h = key 16 s + 8 (j >> 2) + 4 h + (j & 3)),
//     P = p * 2^14 split into hi + lo; the V fragment (B operand, the same key order) comes out of row-major V by two
//     ds_read_b64_tr_b16 (4 keys x 16 columns per 16-lane group, delivered column-major), V chunks swizzled by
//     ((row >> 1) & 1) << 2 so that the four rows of a transposed read sit on disjoint banks;
//   * O = acc 2^-14 is already in the window's scale: it leaves as the plane pair of the out-projection's A operand.
// Matrix cycles: 192 MFMAs of 32 cycles per wave against 512 of 64 on the fp32 pipe.
using ah8 = __attribute__((ext_vector_type(8))) _Float16;
typedef short atr4 __attribute__((__vector_size__(4 * sizeof(short))));

__global__ __launch_bounds__(256, 2) void attn_f16x3_kernel(const unsigned short* __restrict__ qkvp, int64_t in_plane,
                                                            unsigned short* __restrict__ planes, int64_t plane_stride,
                                                            int64_t n_rows, const int* __restrict__ Tw,
                                                            const int64_t* __restrict__ row0, int NH, int Hd, float scale,
                                                            const float* __restrict__ row_scale) {
    constexpr int HD = 64, KB = 128, PLANE = KB * HD, TILE = 2 * PLANE;          // one staged block: 2 planes x 16 KB (halfs)
    extern __shared__ __attribute__((aligned(1024))) unsigned short kvh[];      // two blocks
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int win = blockIdx.x / NH, head = blockIdx.x - win * NH;
    const int T = Tw[win];
    if (blockIdx.y * 128 >= T) return;                                          // (workgroup-uniform)
    const int64_t wrow0 = row0[win];
    const int64_t ld = 3 * (int64_t)Hd;                                         // halfs per row of a plane
    const unsigned short* base = qkvp + wrow0 * ld + (int64_t)head * HD;       // q of this window and head, plane 0
    const int q0 = blockIdx.y * 128 + wv * 32;
    const int nblk = (T + KB - 1) / KB;                                         // 1 or 2 key blocks
    const float sw = row_scale[wrow0];                                          // the window's power of two
    const float sinv = pow2_inverse(sw);

    // Q fragments (B operand of S^T = K Q^T): lane (query l31, half h) holds d = 16 s + 8 h + j, both planes
    ah8 qh[4], ql[4];
    {
        const int q = q0 + l31 < T ? q0 + l31 : T - 1;
        const unsigned short* qp = base + (int64_t)q * ld + 8 * h;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            qh[s4] = *reinterpret_cast<const ah8*>(qp + 16 * s4);
            ql[s4] = *reinterpret_cast<const ah8*>(qp + in_plane + 16 * s4);
        }
    }
    af32x16 sc[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[kt][e] = 0.0f;

    // LDS-DMA staging of a 128-key block of K (col0 = Hd) or V (col0 = 2 Hd): one wave-instruction writes 8 rows of 128 B of
    // one plane linearly; chunk c' of row r is fetched from source chunk c' ^ f(r).  Rows past the window re-read its last row.
    auto stage = [&](int col0, int key0, unsigned short* dst, bool is_v) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ins = wv * 8 + i;                                         // 32 wave-instructions per block
            const int pl = ins >> 4, r = 8 * (ins & 15) + (lane >> 3), pc = lane & 7;
            const int key = key0 + r < T ? key0 + r : T - 1;
            const int c = pc ^ (is_v ? (((r >> 1) & 1) << 2) : ((r >> 1) & 7));
            __builtin_amdgcn_global_load_lds((attn_glb_ptr)(base + col0 + pl * in_plane + (int64_t)key * ld + 8 * c),
                                             (attn_lds_ptr)(dst + ins * 512), 16, 0, 0);
        }
    };
    auto drain = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    unsigned short* buf0 = kvh;
    unsigned short* buf1 = kvh + TILE;

    // ---- scores: block b of K lives in buf[b]; the next block (K1, then V0) is in flight during the multiply ----
    stage(Hd, 0, buf0, false);
    drain();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b < nblk) {
            const unsigned short* cur = b == 0 ? buf0 : buf1;
            if (b + 1 < nblk) stage(Hd, KB * (b + 1), buf1, false);            // K1 -> buf1 while K0 is multiplied
            else stage(2 * Hd, 0, b == 0 ? buf1 : buf0, true);                 // last K block: V0 -> the other buffer
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int kt = 4 * b + t4;
                const int row = 32 * t4 + l31;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int off = row * HD + 8 * ((2 * s4 + h) ^ ((row >> 1) & 7));
                    const ah8 kh = *reinterpret_cast<const ah8*>(cur + off);
                    const ah8 kl = *reinterpret_cast<const ah8*>(cur + PLANE + off);
                    sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s4], sc[kt], 0, 0, 0);
                    sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s4], sc[kt], 0, 0, 0);
                    sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s4], sc[kt], 0, 0, 0);
                }
            }
            drain();
        }
    }
    // V0 now sits in buf1 (one key block) or buf0 (two key blocks)
    // ---- softmax over the keys of this lane's query: key = 32 kt + (e & 3) + 8 (e >> 2) + 4 h ----
    const float sscale = scale * sinv * sinv;                                   // q and k both carry s_w
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = 32 * kt + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float v = key < T ? sc[kt][e] * sscale : -INFINITY;
            sc[kt][e] = v;
            m = fmaxf(m, v);
        }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.0f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float ev = expf(sc[kt][e] - m);                               // exp(-inf) = 0 for the padded keys
            sc[kt][e] = ev;
            sum += ev;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 16384.0f / sum;                                           // probabilities travel as p * 2^14 (hi + lo)

    // ---- O = P V ----
    af32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.0f; o1[e] = 0.0f; }
    // transposed-read addressing: 16-lane group gq covers columns 16 (gq & 1) .. + 15 of a column tile and the 4 keys of lane
    // half h = gq >> 1; lane 4 q + p of the group supplies row q, columns 4 p .. 4 p + 3
    const int gq = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b < nblk) {
            // V block b: V0 is where the score phase left it, V1 goes to the other buffer while V0 is multiplied
            unsigned short* v0buf = nblk == 1 ? buf1 : buf0;
            const unsigned short* cur = b == 0 ? v0buf : (v0buf == buf0 ? buf1 : buf0);
            if (b == 0 && nblk > 1) stage(2 * Hd, KB, v0buf == buf0 ? buf1 : buf0, true);
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int kt = 4 * b + t4;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    ah8 ph, pl;                                                 // A operand: this lane's query, keys of k-step s2
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float pn = sc[kt][8 * s2 + j] * inv;
                        const _Float16 hi = (_Float16)pn;
                        ph[j] = hi;
                        pl[j] = (_Float16)(pn - (float)hi);
                    }
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        // (volatile asm with a memory clobber: the reads stay behind the barrier that publishes the staged
                        // block, the wait for the four results is explicit and tied to them)
                        atr4 th0, th1, tl0, tl1;
                        {
                            const int rowa = 32 * t4 + 16 * s2 + 4 * (gq >> 1) + tq, rowb = rowa + 8;      // keys within the staged block
                            const int cha = (4 * ct + 2 * (gq & 1) + (tp >> 1)) ^ (((rowa >> 1) & 1) << 2);
                            const int chb = (4 * ct + 2 * (gq & 1) + (tp >> 1)) ^ (((rowb >> 1) & 1) << 2);
                            const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)(cur + rowa * HD + 8 * cha + 4 * (tp & 1));
                            const unsigned a1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)(cur + rowb * HD + 8 * chb + 4 * (tp & 1));
                            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(th0) : "v"(a0) : "memory");
                            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(th1) : "v"(a1) : "memory");
                            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tl0) : "v"(a0), "n"(2 * PLANE) : "memory");
                            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tl1) : "v"(a1), "n"(2 * PLANE) : "memory");
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(th0), "+v"(th1), "+v"(tl0), "+v"(tl1) : : "memory");
                        }
                        // (whole-vector moves: rebuilding the fragments element by element from the 4-element results,
                        // `vh[j] = bit_cast(th0[j])`, came out of hipcc 7.2 as a broadcast of element 0)
                        typedef short atr8 __attribute__((__vector_size__(8 * sizeof(short))));
                        const ah8 vh = __builtin_bit_cast(ah8, __builtin_shufflevector(th0, th1, 0, 1, 2, 3, 4, 5, 6, 7));
                        const ah8 vl = __builtin_bit_cast(ah8, __builtin_shufflevector(tl0, tl1, 0, 1, 2, 3, 4, 5, 6, 7));
                        if (ct == 0) {
                            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, o0, 0, 0, 0);
                            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, o0, 0, 0, 0);
                            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, o0, 0, 0, 0);
                        } else {
                            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, o1, 0, 0, 0);
                            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, o1, 0, 0, 0);
                            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, o1, 0, 0, 0);
                        }
                    }
                }
            }
            if (b == 0 && nblk > 1) drain();
        }
    }
    // C layout: column = lane & 31 (d), row = (e & 3) + 8 (e >> 2) + 4 h (query within the wave's 32).  acc * 2^-14 = o * s_w:
    // already the out-projection's A operand in the window's scale (|o| <= max |v| of the window, and v * s_w < 2^15).
    // k16 panels of n_rows rows (gemm_f16x3.h): column head * 64 + 32 u + l31 -> panel 4 head + 2 u + (l31 >> 4), k = l31 & 15
    const int64_t panel_sz = n_rows * 16;
    unsigned short* op = planes + ((int64_t)(head * 4) + (l31 >> 4)) * panel_sz + wrow0 * 16 + (l31 & 15);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int q = q0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (q < T) {
            unsigned short* d0 = op + (int64_t)q * 16;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float x = (u == 0 ? o0[e] : o1[e]) * (1.0f / 16384.0f);
                unsigned short a2, b2;
                split2_w(x, a2, b2);
                d0[2 * u * panel_sz] = a2;
                d0[plane_stride + 2 * u * panel_sz] = b2;
            }
        }
    }
}

// one power of two per window for the q / k / v projection's plane output: the bound of the window's rows,
// max_rows |y|_2 * max_n |w_n|_2 + max |b|, written per row (the GEMM's c_scale / the attention's scale / the out-projection's a_scale)
__global__ __launch_bounds__(256) void qkv_scale_kernel(const unsigned* __restrict__ win_norm, const int* __restrict__ rowwin,
                                                        int64_t rows, const unsigned* __restrict__ wst, const unsigned* __restrict__ bst,
                                                        float* __restrict__ row_scale) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float bound = __uint_as_float(win_norm[rowwin[r]]) * __uint_as_float(wst[0]) * 1.000001f + __uint_as_float(bst[1]);
    row_scale[r] = f16x2_scale_for_bound(bound);
}

// ---- row softmax in place, one wave per row of length T (row stride Tp, pad columns zeroed) ---------
// Tp <= 256 (every Wav2Vec2 window: T <= 249): the row lives in one float4 per lane, one load + one store
template <bool SMALL>
__global__ __launch_bounds__(256) void softmax_kernel(float* __restrict__ S, int64_t rows, int T, int Tp) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* p = S + row * Tp;
    if (SMALL) {
        const int c0 = 4 * lane;
        float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (c0 < Tp) v = reinterpret_cast<const float4*>(p)[lane];
        float e[4] = {v.x, v.y, v.z, v.w};
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (c0 + i >= T) e[i] = -INFINITY; m = fmaxf(m, e[i]); }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { e[i] = (c0 + i < T) ? expf(e[i] - m) : 0.f; s += e[i]; }
        s = wave_sum(s);
        const float inv = 1.0f / s;
        if (c0 < Tp) reinterpret_cast<float4*>(p)[lane] = make_float4(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv);
        return;
    }
    float m = -INFINITY;
    for (int c = lane; c < T; c += 64) m = fmaxf(m, p[c]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int c = lane; c < T; c += 64) { const float e = expf(p[c] - m); p[c] = e; s += e; }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int c = lane; c < Tp; c += 64) p[c] = c < T ? p[c] * inv : 0.0f;
}

// ---- regroup for the grouped positional conv: xg[chunk][g][tt][ci], zero padded in time -----------
__global__ __launch_bounds__(256) void regroup_kernel(const float4* __restrict__ x, float4* __restrict__ xg, int n,
                                                      int T, int Hd4, int G, int K) {
    const int cg4 = Hd4 / G;
    const int TT = T + K - 1;
    const int64_t total = (int64_t)n * G * TT * cg4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ci = (int)(i % cg4);
        int64_t r = i / cg4;
        const int tt = (int)(r % TT); r /= TT;
        const int g = (int)(r % G);
        const int64_t chunk = r / G;
        const int t = tt - K / 2;
        xg[i] = (t >= 0 && t < T) ? x[(chunk * T + t) * Hd4 + g * cg4 + ci] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// the same regrouping as two fp16 planes (A operand of the positional convolution on the f16x3 GEMM), scaled per window by
// the power of two that the window's largest |x| (reported by the feature projection's epilogue) asks for
__global__ __launch_bounds__(256) void regroup_planes_kernel(const float4* __restrict__ x, unsigned short* __restrict__ xg,
                                                             int64_t plane, int n, int Tmax, int Hd4, int G, int K,
                                                             const int* __restrict__ Tw, const int64_t* __restrict__ row0,
                                                             const unsigned* __restrict__ amax, float* __restrict__ win_scale) {
    // destination per (window, group): Tmax + K - 1 rows of cg channels, frame t of window w at tt = t + K / 2, zeros around its T_w frames
    const int cg4 = Hd4 / G;
    const int TT = Tmax + K - 1;
    const int64_t total = (int64_t)n * G * TT * cg4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ci = (int)(i % cg4);
        int64_t r = i / cg4;
        const int tt = (int)(r % TT); r /= TT;
        const int g = (int)(r % G);
        const int64_t chunk = r / G;
        const int t = tt - K / 2;
        const float sc = f16x2_scale_for_bound(__uint_as_float(amax[chunk]));
        if (g == 0 && tt == 0 && ci == 0) win_scale[chunk] = sc;
        const float4 v = (t >= 0 && t < Tw[chunk]) ? x[(row0[chunk] + t) * Hd4 + g * cg4 + ci] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float vv[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
        unsigned short hh[4], ll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) split2_w(vv[k], hh[k], ll[k]);
        // k16 panels per (window, group): [cg / 16][TT][16] - tap k of output frame t is panel row t + k, so the GEMM's DMA
        // moves 32 consecutive rows of a panel as one contiguous KiB (a_tap_panels = cg / 16)
        unsigned short* pp = xg + ((((chunk * G + g) * (cg4 / 4) + (ci >> 2)) * TT + tt) * 16 + (ci & 3) * 4);
        *reinterpret_cast<uint2*>(pp) = make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
        *reinterpret_cast<uint2*>(pp + plane) = make_uint2(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16));
    }
}

// ---- positional convolution: the window's panel image stays in LDS -----------------------------------------------------------
// As a batched GEMM (gemm_f16x3, 256 x 64 tiles) every k-tile of the 128-tap convolution re-reads its A tile from L2 although
// consecutive taps differ by ONE ROW: 6.3 MB of DMA per tile, 755 GB per 1 000 clips, 6.7 TB/s - the kernel sat on the L2 -> LDS
// path at 125 TFLOP/s-equivalent whatever the tile shape.  Here a workgroup owns one (window, group): the group's cg channels of
// the zero-padded window ([cg / 16 panels][T_w + PK - 1 rows][16], both fp16 planes: 120 KB at base geometry) are fetched ONCE,
// tap k of output row t is LDS row t + k, and only the weights stream (6 KB per step of two k16 units, four stages).  Eight
// waves x 64 rows; v_mfma_f32_16x16x32_f16 so that the 48 output channels are three full column tiles; the arithmetic is the
// GEMM's (a_l b_h + a_h b_l + a_h b_h, fp32 accumulation, power-of-two scales undone in the epilogue, bias, GELU).
// Rows and halves are stored with the GEMM's swizzle (16-byte half h of row r in slot h ^ ((r >> 3) & 1)): lanes r and r + 8 of a
// fragment read then hit different banks for any tap offset.
typedef _Float16 pc_f16x8 __attribute__((ext_vector_type(8)));
typedef float pc_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* pc_lds_ptr;
typedef const __attribute__((address_space(1))) void* pc_glb_ptr;
constexpr int PC_ROWS = 512;                     // output rows per workgroup with SPLIT = 1 (8 waves x 4 row tiles of 16)

// SPLIT = 2 (windows of at most 256 frames - the 5 s windows of the extractor give 249): 256 rows per workgroup, waves 0-3 take
// the even steps and waves 4-7 the odd ones (split K; the halves meet through LDS in front of the epilogue), so that all eight
// waves have rows to work on and each weight step is read from LDS by four waves instead of eight.
template <int NT, int SPLIT>
__global__ __launch_bounds__(512, 1) void posconv_f16x3_kernel(const unsigned short* __restrict__ xg, int64_t a_plane,
                                                               const unsigned short* __restrict__ wp, int64_t b_plane, int Hd,
                                                               const float* __restrict__ a_scale, const float* __restrict__ b_scale,
                                                               const float* __restrict__ bias, const int* __restrict__ Tw,
                                                               const int64_t* __restrict__ row0, float* __restrict__ y, int G, int TT,
                                                               int PK, int rows_alloc) {
    extern __shared__ __attribute__((aligned(1024))) unsigned short pc_sm[];
    constexpr int cg = 16 * NT, B_SEG = cg * 16, B_STAGE = 4 * B_SEG, B_INSTR = cg / 8, RGRPS = 8 / SPLIT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = blockIdx.x / G, g = blockIdx.x - w * G;
    const int a_panel = rows_alloc * 16;                                   // elements per (plane, panel)
    unsigned short* sA = pc_sm;
    unsigned short* sB = pc_sm + 2 * NT * a_panel;
    // the window's image -> LDS: instruction ii = (plane, panel, block of 32 rows), 1 KiB each
    {
        const int blocks = rows_alloc / 32, total = 2 * NT * blocks;
        const int r_in = lane >> 1, slot = lane & 1;
        for (int ii = wave; ii < total; ii += 8) {
            const int pp = ii / blocks, rb = ii - pp * blocks;             // pp = plane * NT + panel
            const int pl = pp / NT, panel = pp - pl * NT;
            const int row = rb * 32 + r_in;
            const int srow = row < TT ? row : TT - 1;
            const int chunk = slot ^ ((row >> 3) & 1);
            const unsigned short* src = xg + pl * a_plane + ((((int64_t)w * G + g) * NT + panel) * TT + srow) * 16 + chunk * 8;
            __builtin_amdgcn_global_load_lds((pc_glb_ptr)src, (pc_lds_ptr)(sA + pp * a_panel + rb * 512), 16, 0, 0);
        }
    }
    // weights: step j = units 2 j, 2 j + 1 (a unit = 16 k of one tap); stage layout [unit in pair][plane][cg rows][16].
    // Iteration i = steps SPLIT i .. SPLIT i + SPLIT - 1, ring of four iterations.
    const int nit = PK * NT / 2 / SPLIT;
    int64_t boff = 0;
    if (wave < B_INSTR) {
        const int c = wave * 64 + lane;                                    // 16-byte chunk of the stage
        const int seg = c / (2 * cg), within = c - seg * (2 * cg);
        const int row = within >> 1, slot = within & 1;
        boff = (int64_t)(seg & 1) * b_plane + ((int64_t)(seg >> 1) * Hd + (g * cg + row)) * 16 + (slot ^ ((row >> 3) & 1)) * 8;
    }
    auto b_dma = [&](int i) {
        if (wave < B_INSTR) {
#pragma unroll
            for (int kh = 0; kh < SPLIT; ++kh)
                __builtin_amdgcn_global_load_lds((pc_glb_ptr)(wp + boff + (int64_t)(SPLIT * i + kh) * 2 * Hd * 16),
                                                 (pc_lds_ptr)(sB + ((i & 3) * SPLIT + kh) * B_STAGE + wave * 512), 16, 0, 0);
        }
    };
    b_dma(0);
    if (nit > 1) b_dma(1);
    if (nit > 2) b_dma(2);
    pc_f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = pc_f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, q = lane >> 4, hlf = q & 1, upair = q >> 1;
    const int khalf = wave / RGRPS, m_base = (wave - khalf * RGRPS) * 64;
    // fragments of a step: B from its stage of the ring, A from the resident image at the step's tap
    struct Frag { pc_f16x8 ah[4], al[4], bh[NT], bl[NT]; };
    auto fetch = [&](int i, Frag& F) {
        const int unit = 2 * (SPLIT * i + khalf) + upair;
        const int tap = unit / NT, panel = unit - tap * NT;
        const unsigned short* bst = sB + ((i & 3) * SPLIT + khalf) * B_STAGE + upair * 2 * B_SEG;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int brow = 16 * nt + r16;
            const int off = brow * 16 + ((hlf ^ ((brow >> 3) & 1)) << 3);
            F.bh[nt] = *reinterpret_cast<const pc_f16x8*>(bst + off);
            F.bl[nt] = *reinterpret_cast<const pc_f16x8*>(bst + B_SEG + off);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int row = m_base + 16 * mt + r16 + tap;
            const int off = panel * a_panel + row * 16 + ((hlf ^ ((row >> 3) & 1)) << 3);
            F.ah[mt] = *reinterpret_cast<const pc_f16x8*>(sA + off);
            F.al[mt] = *reinterpret_cast<const pc_f16x8*>(sA + NT * a_panel + off);
        }
    };
    auto mfmas = [&](const Frag& F) {                                     // product by product: 4 NT independent accumulators
#pragma unroll                                                             // between two instructions on the same one
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F.al[mt], F.bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F.ah[mt], F.bl[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F.ah[mt], F.bh[nt], acc[mt][nt], 0, 0, 0);
    };
    // Software pipeline: at the top of iteration i the fragments of iteration i are in registers; the wave waits for its share
    // of iteration i + 1's weights (issued two iterations ago), the barrier publishes everybody's and frees the ring slot of
    // iteration i + 3 (read as fragments during iteration i - 2), the DMA of iteration i + 3 goes out, the fragments of
    // iteration i + 1 are requested and the 36 MFMAs of this iteration run while they arrive.
    Frag F0, F1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the image and the first three weight iterations
    __syncthreads();
    fetch(0, F0);
    auto step = [&](int i, const Frag& cur, Frag& nxt) {
        if (i >= 1) {
            if (i + 2 < nit) { if constexpr (SPLIT == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (i + 3 < nit) b_dma(i + 3);
        if (i + 1 < nit) fetch(i + 1, nxt);
        mfmas(cur);
    };
    for (int i = 0; i < nit; i += 2) {                                     // nit is even (launcher)
        step(i, F0, F1);
        step(i + 1, F1, F0);
    }
    if constexpr (SPLIT == 2) {                                            // the odd steps' sums join the even steps' through LDS
        __syncthreads();                                                   // (the image is dead)
        float* red = reinterpret_cast<float*>(pc_sm) + (wave - khalf * RGRPS) * (4 * NT * 4 * 64);
        if (khalf == 1) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int v = 0; v < 4; ++v) red[((mt * NT + nt) * 4 + v) * 64 + lane] = acc[mt][nt][v];
        }
        __syncthreads();
        if (khalf == 1) return;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[mt][nt][v] += red[((mt * NT + nt) * 4 + v) * 64 + lane];
    }
    // epilogue: D of the 16 x 16 tile: lane (q, r16), register v = row 4 q + v, column r16
    const int Tv = Tw[w];
    const int64_t r0 = row0[w];
    const float sa_inv = pow2_inverse(a_scale[w]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = g * cg + 16 * nt + r16;
        const float inv = sa_inv * pow2_inverse(b_scale[col]);
        const float bs = bias[col];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int v = 0; v < 4; v += 2) {
                const int m = m_base + 16 * mt + 4 * q + v;
                const gelu_f32x2 g2 = gelu_pair(gelu_f32x2{acc[mt][nt][v] * inv + bs, acc[mt][nt][v + 1] * inv + bs});
                if (m < Tv) y[(r0 + m) * Hd + col] = g2.x;
                if (m + 1 < Tv) y[(r0 + m + 1) * Hd + col] = g2.y;
            }
    }
}

static int posconv_rows_alloc(int split, int PK) { return (PC_ROWS / split + PK - 1 + 31) & ~31; }
static size_t posconv_lds_bytes(int cg, int PK, int split) {
    return ((size_t)2 * (cg / 16) * posconv_rows_alloc(split, PK) * 16 + (size_t)4 * split * 4 * cg * 16) * sizeof(unsigned short);
}

static int ln(const float* x, const float* r, const float* g, const float* b, float* out, int64_t rows, int D,
              float eps, hipStream_t s, const int64_t* out_row_start = nullptr, const int* rowwin = nullptr,
              const int64_t* row0 = nullptr, unsigned short* planes = nullptr, bool panel = false, float* scale_out = nullptr,
              const unsigned* bound_w = nullptr, const unsigned* bound_b = nullptr, float* bound_scale_out = nullptr,
              unsigned* win_norm = nullptr) {
    const int64_t blocks = (rows + 3) / 4;
    RSAF_CHECK_ARG(blocks <= 0x7fffffffLL, "too many rows");
    RSAF_CHECK_ARG(!planes || scale_out, "planes need their scale array");
    RSAF_CHECK_ARG(!win_norm || rowwin, "the per-window norm needs the row -> window map");
    ProfScope prof("w2v2_layernorm", s, 0.0, (double)rows * D * (4 * (r ? 2 : 1) + (out ? 4 : 0) + (planes ? 4 : 0)));
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, r, g, b, out, rows, D, eps,
                       out_row_start, rowwin, row0, planes, rows * D, panel ? 1 : 0, scale_out, bound_w, bound_b, bound_scale_out, win_norm);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

template <bool APPLY>
static int conv0_launch(const Cfg& c, const float* xn, const float* w0, float* part, const float* ab, const float* scale,
                        unsigned short* outp, int64_t plane, int n, int len, const int* T0w, int T0, int slab, int slabs, hipStream_t s) {
    const int threads = c.C <= 256 ? c.C : 256;
    const int cpt = c.C / threads;
    dim3 grid((unsigned)slabs, (unsigned)n);
    // HBM-bound: the apply pass writes C channels x T0 frames per window as two fp16 planes (4 B per element: 33 MB per
    // 5 s window at C = 512) and reads the window once; the statistics pass only reads the window (both recompute the
    // 10-tap convolution: Cin = 1, 0.16 GFLOP per window)
    ProfScope prof(APPLY ? "w2v2_conv0_apply" : "w2v2_conv0_stats", s, 0.0,
                   APPLY ? (double)n * ((double)c.C * T0 * 4.0 + 4.0 * (5.0 * T0 + 5.0)) : (double)n * 4.0 * (5.0 * T0 + 5.0));
#define RSAF_C0(CPT)                                                                                       \
    hipLaunchKernelGGL((conv0_kernel<CPT, APPLY>), grid, dim3(threads), 0, s, xn, w0, part, ab, scale, outp, plane, len, T0w, T0, \
                       c.C, slab, slabs)
    switch (cpt) {
        case 1: RSAF_C0(1); break;
        case 2: RSAF_C0(2); break;
        case 3: RSAF_C0(3); break;
        case 4: RSAF_C0(4); break;
        default: set_error("conv0: unsupported conv_dim"); return RSAF_ERR_ARG;
    }
#undef RSAF_C0
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

}  // namespace w2v2
}  // namespace rsaf

using namespace rsaf;
using namespace rsaf::w2v2;

extern "C" {

int rsaf_w2v2_frames(int chunk_len) {
    int T[7];
    chunk_lengths(chunk_len, T);
    return T[6];
}

int64_t rsaf_w2v2_weight_floats(int conv_dim, int hidden, int layers, int heads, int intermediate, int pos_kernel,
                                int pos_groups) {
    Cfg c{conv_dim, hidden, layers, heads, intermediate, pos_kernel, pos_groups, 1e-5f};
    if (check_cfg(c) != RSAF_OK) return -1;
    return make_layout(c).total;
}

int rsaf_w2v2_weight_offsets(int conv_dim, int hidden, int layers, int heads, int intermediate, int pos_kernel,
                             int pos_groups, int64_t* offsets_host, int cap, int* n_host) {
    Cfg c{conv_dim, hidden, layers, heads, intermediate, pos_kernel, pos_groups, 1e-5f};
    int rc = check_cfg(c);
    if (rc != RSAF_OK) return rc;
    RSAF_CHECK_ARG(offsets_host && n_host, "NULL output");
    const Layout L = make_layout(c);
    std::vector<int64_t> v = {L.conv0, L.gng, L.gnb};
    for (int i = 0; i < 6; ++i) v.push_back(L.conv[i]);
    for (int64_t o : {L.fplg, L.fplb, L.fpw, L.fpb, L.posw, L.posb, L.elng, L.elnb}) v.push_back(o);
    for (const LayerOff& lo : L.layers)
        for (int64_t o : {lo.wqkv, lo.bqkv, lo.wo, lo.bo, lo.ln1g, lo.ln1b, lo.w1, lo.b1, lo.w2, lo.b2, lo.ln2g, lo.ln2b})
            v.push_back(o);
    RSAF_CHECK_ARG(cap >= (int)v.size(), "offsets_host too small");
    for (size_t i = 0; i < v.size(); ++i) offsets_host[i] = v[i];
    *n_host = (int)v.size();
    return RSAF_OK;
}

static int forward_impl(const float* wav, const int64_t* chunk_start, const int* len_dev_or_null, const Rag& R, const Cfg& c,
                        const float* weights, void* workspace, int64_t workspace_bytes, float* out, const int64_t* out_row_start,
                        hipStream_t s);

int64_t rsaf_w2v2_workspace_bytes(int n_chunks, int chunk_len, int conv_dim, int hidden, int layers, int heads,
                                  int intermediate, int pos_kernel, int pos_groups) {
    Cfg c{conv_dim, hidden, layers, heads, intermediate, pos_kernel, pos_groups, 1e-5f};
    if (check_cfg(c) != RSAF_OK || n_chunks <= 0) return -1;
    if (rsaf_w2v2_frames(chunk_len) <= 0) return -1;
    std::vector<int> len((size_t)n_chunks, chunk_len);
    Rag R;
    if (make_rag(len.data(), n_chunks, R) != RSAF_OK) return -1;
    return make_ws(c, R).total * (int64_t)sizeof(float);
}

int64_t rsaf_w2v2_workspace_bytes_ragged(const int* chunk_len_host, int n_chunks, int conv_dim, int hidden, int layers, int heads,
                                         int intermediate, int pos_kernel, int pos_groups) {
    Cfg c{conv_dim, hidden, layers, heads, intermediate, pos_kernel, pos_groups, 1e-5f};
    if (check_cfg(c) != RSAF_OK || n_chunks <= 0 || !chunk_len_host) return -1;
    Rag R;
    if (make_rag(chunk_len_host, n_chunks, R) != RSAF_OK) return -1;
    return make_ws(c, R).total * (int64_t)sizeof(float);
}

int rsaf_w2v2_forward(const float* wav, const int64_t* chunk_start, int n_chunks, int chunk_len, int conv_dim,
                      int hidden, int layers, int heads, int intermediate, int pos_kernel, int pos_groups,
                      float layer_norm_eps, const float* weights, void* workspace, int64_t workspace_bytes,
                      float* out, const int64_t* out_row_start, rsaf_stream_t stream) {
    Cfg c{conv_dim, hidden, layers, heads, intermediate, pos_kernel, pos_groups, layer_norm_eps};
    int rc = check_cfg(c);
    if (rc != RSAF_OK) return rc;
    RSAF_CHECK_ARG(n_chunks >= 0, "negative chunk count");
    if (n_chunks == 0) return RSAF_OK;
    std::vector<int> len((size_t)n_chunks, chunk_len);
    Rag R;
    if ((rc = make_rag(len.data(), n_chunks, R))) return rc;
    return forward_impl(wav, chunk_start, nullptr, R, c, weights, workspace, workspace_bytes, out, out_row_start, (hipStream_t)stream);
}

int rsaf_w2v2_forward_ragged(const float* wav, const int64_t* chunk_start, const int* chunk_len, const int* chunk_len_host,
                             int n_chunks, int conv_dim, int hidden, int layers, int heads, int intermediate, int pos_kernel,
                             int pos_groups, float layer_norm_eps, const float* weights, void* workspace,
                             int64_t workspace_bytes, float* out, const int64_t* out_row_start, rsaf_stream_t stream) {
    Cfg c{conv_dim, hidden, layers, heads, intermediate, pos_kernel, pos_groups, layer_norm_eps};
    int rc = check_cfg(c);
    if (rc != RSAF_OK) return rc;
    RSAF_CHECK_ARG(n_chunks >= 0, "negative chunk count");
    if (n_chunks == 0) return RSAF_OK;
    RSAF_CHECK_ARG(chunk_len && chunk_len_host, "NULL length table");
    Rag R;
    if ((rc = make_rag(chunk_len_host, n_chunks, R))) return rc;
    return forward_impl(wav, chunk_start, chunk_len, R, c, weights, workspace, workspace_bytes, out, out_row_start, (hipStream_t)stream);
}

}  // extern "C"

static int forward_impl(const float* wav, const int64_t* chunk_start, const int* len_dev_or_null, const Rag& R, const Cfg& c,
                        const float* weights, void* workspace, int64_t workspace_bytes, float* out, const int64_t* out_row_start,
                        hipStream_t s) {
    int rc = RSAF_OK;
    const int n_chunks = R.n;
    RSAF_CHECK_ARG(n_chunks <= 65535 / std::max(c.NH, c.PG), "too many chunks per call");
    RSAF_CHECK_ARG(wav && chunk_start && weights && workspace && out, "NULL pointer");
    const int* T = R.Tmax;                                   // frames of the LONGEST window per layer (strides, grids)
    const Workspace W = make_ws(c, R);
    if (workspace_bytes < W.total * (int64_t)sizeof(float)) {
        set_error("rsaf_w2v2_forward: workspace too small");
        return RSAF_ERR_WORKSPACE;
    }
    const Layout L = make_layout(c);
    float* ws = static_cast<float*>(workspace);
    const float* Wt = weights;
    const int n = n_chunks, C = c.C, Hd = c.Hd, Tt = T[6];
    const int64_t rows = R.rows;
    RSAF_CHECK_ARG(rows <= 0x7fffffffLL, "too many frames per call");

    auto planes_at = [&](int64_t off) { return reinterpret_cast<uint16_t*>(ws + off); };
    auto bits_at = [&](int64_t off) { return reinterpret_cast<unsigned*>(ws + off); };
    auto wstat = [&](int idx) { return bits_at(W.wstat) + 2 * idx; };        // {max row norm, max |element|} of matrix idx
    // window tables on the device (every window has its own length: the reference's tail windows)
    int* Tw = reinterpret_cast<int*>(ws + W.t_Tw);           // [7][n]
    int64_t* row0 = reinterpret_cast<int64_t*>(ws + W.t_row0);
    int64_t* ztab = reinterpret_cast<int64_t*>(ws + W.t_ztab);   // [7][n][2]
    int* rowwin = reinterpret_cast<int*>(ws + W.t_rowwin);
    const int* wlen = len_dev_or_null;
    {
        if (!wlen) {                                         // equal windows: the length table is filled here
            int* wl = reinterpret_cast<int*>(ws + W.wlen);
            hipLaunchKernelGGL(fill_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, wl, n, R.maxlen);
            wlen = wl;
        }
        hipLaunchKernelGGL(w2v2_tables_kernel, dim3(1), dim3(256), 0, s, wlen, n, C, Hd, Tw, row0, ztab);
        hipLaunchKernelGGL(w2v2_rowwin_kernel, dim3(n), dim3(256), 0, s, row0, rowwin);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    // helper: C = act(A B^T + bias (+ R)) on the f16x3 kernel; A / B as fp16 plane pairs with their scales
    struct Out { float* Cf; int64_t sC; uint16_t* Cp; int64_t c_plane, sCp; const float* c_scale; int cs_zs, cs_ms; bool cp_panel; };
    auto gemm3 = [&](const uint16_t* A, int64_t a_plane, int64_t lda, int64_t sA, const float* a_scale, int as_zs, int as_ms,
                     const uint16_t* B, const float* b_scale, int M, int N, int K, const Out& o, const float* bias, const float* Rr,
                     int nz, int act, const char* tag, bool a_panel, unsigned* amax = nullptr, int amax_zs = 0,
                     const int* amax_row_slot = nullptr, int amax_col_min = 0, const int64_t* ztab = nullptr) {
        GemmH3Params p{};
        p.a_panel = a_panel; p.b_panel = 1; p.cp_panel = o.cp_panel;
        p.A = A; p.a_plane = a_plane; p.lda = lda; p.sA = sA; p.a_scale = a_scale; p.a_scale_zs = as_zs; p.a_scale_ms = as_ms;
        p.B = B; p.b_plane = (int64_t)N * K; p.ldb = 16; p.b_scale = b_scale;
        p.C = o.Cf; p.ldc = N; p.sC = o.sC;
        p.Cp = o.Cp; p.c_plane = o.c_plane; p.ldcp = N; p.sCp = o.sCp; p.c_scale = o.c_scale; p.c_scale_zs = o.cs_zs; p.c_scale_ms = o.cs_ms;
        p.amax_out = amax; p.amax_zs = amax_zs; p.amax_row_slot = amax_row_slot; p.amax_col_min = amax_col_min;
        p.bias = bias; p.R = Rr; p.ldr = N; p.sR = o.sC;
        p.M = M; p.N = N; p.K = K; p.nz = nz; p.ztab = ztab; p.act = act; p.alpha = 1.0f; p.group_m = 0;
        return launch_gemm_f16x3(p, s, tag);
    };
    // 0. weights of the dense layers as fp16 plane pairs in the k16-panel layout, each row with its own power-of-two scale
    //    (once per call: 0.4 GB at base geometry, < 1 ms), and per matrix the largest row norm: the Cauchy-Schwarz factor of
    //    the bound behind the scale of a GEMM's PLANE output (conv1..5, ffn1)
    RSAF_CHECK_HIP(hipMemsetAsync(ws + W.wstat, 0, sizeof(float) * 2 * (WSTAT_LAYER0 + WSTAT_PER_LAYER * c.L), s));
    {
        auto split_wp = [&](int64_t src_off, int64_t nrows, int K, int64_t dst_off, int64_t scale_off, int stat_idx) {
            int r2 = launch_f16x2_row_scales(Wt + src_off, nrows, K, K, ws + scale_off, nullptr, wstat(stat_idx), s);
            if (r2) return r2;
            return launch_split_f16x2(Wt + src_off, nrows, K, K, ws + scale_off, 1, planes_at(dst_off), nrows * K, 1, s);
        };
        for (int i = 0; i < 6; ++i)
            if ((rc = split_wp(L.conv[i], C, KERN[i + 1] * C, W.wp_conv[i], W.ws_conv[i], wstat_conv(i)))) return rc;
        if ((rc = split_wp(L.fpw, Hd, C, W.wp_fp, W.ws_fp, WSTAT_FP))) return rc;
        if ((Hd / c.PG) % 16 == 0) {                       // positional conv on the f16x3 GEMM: [G cg][PK cg] as panels of Hd rows
            if ((rc = split_wp(L.posw, Hd, c.PK * (Hd / c.PG), W.wp_pos, W.ws_pos, WSTAT_POS))) return rc;
        }
        for (int l = 0; l < c.L; ++l) {
            const LayerOff& lo = L.layers[l];
            const int b0 = WSTAT_LAYER0 + WSTAT_PER_LAYER * l;
            if ((rc = split_wp(lo.wqkv, 3 * Hd, Hd, W.wp_qkv[l], W.ws_qkv[l], b0))) return rc;
            if ((rc = split_wp(lo.wo, Hd, Hd, W.wp_o[l], W.ws_o[l], b0 + 1))) return rc;
            if ((rc = split_wp(lo.w1, c.I, Hd, W.wp_1[l], W.ws_1[l], b0 + 2))) return rc;
            if ((rc = split_wp(lo.w2, Hd, c.I, W.wp_2[l], W.ws_2[l], b0 + 3))) return rc;
            // max |b1| (word 1 of the statistics of the bias seen as one row); the scale it writes goes to a scratch slot
            if ((rc = launch_f16x2_row_scales(Wt + lo.b1, 1, c.I, c.I, ws + W.pos_scale, nullptr, wstat(b0 + 4), s))) return rc;
            if ((rc = launch_f16x2_row_scales(Wt + lo.bqkv, 1, 3 * Hd, 3 * Hd, ws + W.pos_scale, nullptr, wstat(b0 + 5), s))) return rc;
        }
    }
    // 1-3. feature encoder, CONV_GROUP windows at a time (its activations are the large ones: 15 999 x 512 per window)
    // window groups of (almost) equal size: ceil(n / G) groups instead of full ones and a small remainder
    const int n_groups = (n + W.G - 1) / W.G, gstep = (n + n_groups - 1) / n_groups;
    float* cscale = ws + W.conv_scale;                       // [7][G]: scale of layer i's plane output, per window of the group
    unsigned* camax = bits_at(W.conv_amax);                  // [7][G]: largest |output| of layer i (layer 0: its bound)
    for (int g0 = 0; g0 < n; g0 += gstep) {
        const int g = std::min(gstep, n - g0);
        // the group's longest window is its first (lengths are non-increasing): its frame counts size the group's launches
        int Tg[7];
        chunk_lengths(R.len[g0], Tg);
        RSAF_CHECK_HIP(hipMemsetAsync(camax, 0, sizeof(unsigned) * 7 * W.G, s));
        // 1. per-chunk normalisation (HF feature extractor)
        {
            ProfScope prof("w2v2_normalize", s, 0.0, (double)g * R.len[g0] * 4 * 3);
            hipLaunchKernelGGL(normalize_kernel, dim3(g), dim3(256), 0, s, wav, chunk_start + g0, wlen + g0, R.maxlen, ws + W.xn);
            RSAF_CHECK_HIP(hipGetLastError());
        }
        // 2. conv0 + GroupNorm + GELU (stats pass, finalize, apply pass); the apply pass writes fp16 plane pairs
        const int slabs_g = (Tg[0] + STAT_SLAB - 1) / STAT_SLAB;
        rc = conv0_launch<false>(c, ws + W.xn, Wt + L.conv0, ws + W.part, nullptr, nullptr, nullptr, 0, g, R.maxlen, Tw + g0, T[0],
                                 STAT_SLAB, slabs_g, s);
        if (rc) return rc;
        {
            const int64_t tot = (int64_t)g * C;
            hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ws + W.part,
                               Wt + L.gng, Wt + L.gnb, ws + W.ab, camax, g, C, slabs_g, Tw + g0);
            RSAF_CHECK_HIP(hipGetLastError());
            if ((rc = launch_scale_from_bound(camax, g, nullptr, 1.0f, nullptr, cscale, s))) return rc;
        }
        {
            const int slab = 128;
            rc = conv0_launch<true>(c, ws + W.xn, Wt + L.conv0, nullptr, ws + W.ab, cscale, planes_at(W.P), (int64_t)W.G * T[0] * C, g,
                                    R.maxlen, Tw + g0, T[0], slab, (Tg[0] + slab - 1) / slab, s);
            if (rc) return rc;
        }
        // 3. conv1..6 as GEMMs over the channels-last sequence (lda = stride * C, K = taps * C) with fused GELU;
        //    the output goes out as planes (the next layer's A), the last one as fp32 rows for the LayerNorm.
        //    Scale of layer i's output, per window: |GELU(x)| <= |x| <= |a|_2 |w|_2 <= sqrt(K) max|a| max_n |w_n|_2 with
        //    max|a| = the largest |output| of layer i - 1, which that layer's epilogue reported (layer 0: the GroupNorm bound).
        //    Every window keeps the longest window's row allotment (batch stride T[i] rows) and has its own row count
        //    (ztab); the last layer writes its rows packed (window w at row row0[w]).
        uint16_t* cur = planes_at(W.P);
        uint16_t* nxt = planes_at(W.Q);
        for (int i = 1; i < 7; ++i) {
            const bool last = i == 6;
            const int K = KERN[i] * C;
            Out o{};
            if (last) { o.Cf = ws + W.c6; o.sC = 0; }
            else {
                if ((rc = launch_scale_from_bound(camax + (int64_t)(i - 1) * W.G, g, reinterpret_cast<const float*>(wstat(wstat_conv(i - 1))),
                                                  sqrtf((float)K) * 1.00001f, nullptr, cscale + (int64_t)i * W.G, s))) return rc;
                o.Cp = nxt; o.c_plane = (int64_t)W.G * T[i] * C; o.sCp = (int64_t)T[i] * C; o.sC = (int64_t)T[i] * C;
                o.c_scale = cscale + (int64_t)i * W.G; o.cs_zs = 1; o.cs_ms = 0;
            }
            rc = gemm3(cur, (int64_t)W.G * T[i - 1] * C, (int64_t)STRD[i] * C, (int64_t)T[i - 1] * C, cscale + (int64_t)(i - 1) * W.G, 1, 0,
                       planes_at(W.wp_conv[i - 1]), ws + W.ws_conv[i - 1], Tg[i], C, K, o, nullptr, nullptr, g, ACT_GELU, "w2v2_gemm",
                       false, last ? nullptr : camax + (int64_t)i * W.G, 1, nullptr, 0, ztab + ((int64_t)(i - 1) * n + g0) * 2);
            if (rc) return rc;
            std::swap(cur, nxt);
        }
    }
    // 4. feature projection: LayerNorm (-> planes, exact row scales) + Linear; its epilogue reports max |x| per window
    RSAF_CHECK_HIP(hipMemsetAsync(ws + W.fp_amax, 0, sizeof(unsigned) * n, s));
    rc = ln(ws + W.c6, nullptr, Wt + L.fplg, Wt + L.fplb, nullptr, rows, C, c.eps, s, nullptr, nullptr, nullptr, planes_at(W.lnfp), true, ws + W.s_lnfp);
    if (rc) return rc;
    {
        Out o{}; o.Cf = ws + W.x;
        rc = gemm3(planes_at(W.lnfp), rows * C, C, 0, ws + W.s_lnfp, 0, 1, planes_at(W.wp_fp), ws + W.ws_fp, (int)rows, Hd, C, o,
                   Wt + L.fpb, nullptr, 1, ACT_NONE, "w2v2_gemm", true, bits_at(W.fp_amax), 0, rowwin);
        if (rc) return rc;
    }
    const int hd = Hd / c.NH;
    const float scale = 1.0f / sqrtf((float)hd);
    static const bool fused_attn = [] { const char* e = getenv("RSAF_W2V2_FUSED_ATTN"); return e ? atoi(e) != 0 : true; }();
    const bool fused = fused_attn && hd == 64 && Tt <= 256;  // attention on the fp16 matrix pipe, q / k / v as plane pairs
    // 5. positional conv embedding (grouped, weight norm folded), GELU, x = LN(x + pos)
    {
        const int cg = Hd / c.PG;
        const int TT = Tt + c.PK - 1;
        const int64_t tot4 = (int64_t)n * TT * (Hd / 4);
        if (cg % 16 == 0) {
            // grouped conv as a two-level batched GEMM on the f16x3 kernel's 256 x 64 tile: batch (window, group), M = T_w rows
            // (the regrouped sequence as k16 panels of T_w + PK - 1 rows: tap k = one row down), N = cg output channels, K = PK cg
            const int64_t plane = (int64_t)n * TT * Hd;
            {
                ProfScope prof("w2v2_regroup", s, 0.0, (double)tot4 * 32);
                hipLaunchKernelGGL(regroup_planes_kernel, dim3((unsigned)std::min<int64_t>((tot4 + 255) / 256, 4096)), dim3(256),
                                   0, s, reinterpret_cast<const float4*>(ws + W.x), reinterpret_cast<unsigned short*>(planes_at(W.xg)),
                                   plane, n, Tt, Hd / 4, c.PG, c.PK, Tw + (int64_t)6 * n, row0, bits_at(W.fp_amax), ws + W.pos_scale);
                RSAF_CHECK_HIP(hipGetLastError());
            }
            const bool pc_off = [] { const char* e = getenv("RSAF_W2V2_POSCONV_GEMM"); return e && e[0] == '1'; }();   // per call: the tests toggle it
            const int split = Tt <= PC_ROWS / 2 ? 2 : 1;
            const bool resident = !pc_off && cg <= 64 && Tt <= PC_ROWS && (c.PK * (cg / 16)) % (4 * split) == 0 &&
                                  posconv_lds_bytes(cg, c.PK, split) <= 160 * 1024;
            if (resident) {
                // the window's image stays in LDS (posconv_f16x3_kernel); RSAF_W2V2_POSCONV_GEMM=1: the batched GEMM below (A/B)
                const int rows_alloc = posconv_rows_alloc(split, c.PK);
                const size_t lds = posconv_lds_bytes(cg, c.PK, split);
                ProfScope prof("w2v2_posconv_gemm", s, 2.0 * (double)rows * cg * (double)c.PK * cg * c.PG, 0.0);
#define RSAF_PC_LAUNCH(NT_, SP_)                                                                                               \
    do {                                                                                                                       \
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)posconv_f16x3_kernel<NT_, SP_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((posconv_f16x3_kernel<NT_, SP_>), dim3((unsigned)(n * c.PG)), dim3(512), lds, s,                    \
                           reinterpret_cast<const unsigned short*>(planes_at(W.xg)), plane,                                    \
                           reinterpret_cast<const unsigned short*>(planes_at(W.wp_pos)), (int64_t)Hd * c.PK * cg, Hd,          \
                           ws + W.pos_scale, ws + W.ws_pos, Wt + L.posb, Tw + (int64_t)6 * n, row0, ws + W.y, c.PG, TT, c.PK,   \
                           rows_alloc);                                                                                        \
    } while (0)
#define RSAF_PC_LAUNCH_NT(SP_)                                                                                                 \
    switch (cg / 16) {                                                                                                         \
        case 1: RSAF_PC_LAUNCH(1, SP_); break;                                                                                 \
        case 2: RSAF_PC_LAUNCH(2, SP_); break;                                                                                 \
        case 3: RSAF_PC_LAUNCH(3, SP_); break;                                                                                 \
        default: RSAF_PC_LAUNCH(4, SP_); break;                                                                                \
    }
                if (split == 2) { RSAF_PC_LAUNCH_NT(2) } else { RSAF_PC_LAUNCH_NT(1) }
#undef RSAF_PC_LAUNCH_NT
#undef RSAF_PC_LAUNCH
                RSAF_CHECK_HIP(hipGetLastError());
            } else {
            GemmH3Params p{};
            p.A = planes_at(W.xg); p.a_plane = plane; p.lda = 16; p.sA = (int64_t)c.PG * TT * cg; p.sA2 = (int64_t)TT * cg;
            p.a_panel = 1; p.a_panel_rows = TT; p.a_tap_panels = cg / 16;
            p.a_scale = ws + W.pos_scale; p.a_scale_zs = 1; p.a_scale_ms = 0;
            p.B = planes_at(W.wp_pos); p.b_plane = (int64_t)Hd * c.PK * cg; p.ldb = 16; p.b_panel = 1; p.b_panel_rows = Hd; p.sB2 = (int64_t)cg * 16;
            p.b_scale = ws + W.ws_pos;
            p.C = ws + W.y; p.ldc = Hd; p.sC = 0; p.sC2 = cg;
            p.bias = Wt + L.posb; p.sBias2 = cg;
            p.M = Tt; p.ztab = ztab + (int64_t)6 * n * 2; p.N = cg; p.K = c.PK * cg; p.nz = n * c.PG; p.nz2 = c.PG; p.act = ACT_GELU; p.alpha = 1.0f;
            rc = launch_gemm_f16x3(p, s, "w2v2_posconv_gemm");
            if (rc) return rc;
            }
        } else {
            // group widths that are no multiple of 16 (test geometries): exact-fp32 GEMM, one launch per run of equal windows
            for (const auto& tg : R.tgroups) {
                const int nw = tg.second - tg.first, Tq = R.T6[tg.first], TTq = Tq + c.PK - 1;
                const int64_t r0 = R.row0[tg.first];
                const int64_t t4 = (int64_t)nw * TTq * (Hd / 4);
                {
                    ProfScope prof("w2v2_regroup", s, 0.0, (double)t4 * 32);
                    hipLaunchKernelGGL(regroup_kernel, dim3((unsigned)std::min<int64_t>((t4 + 255) / 256, 4096)), dim3(256),
                                       0, s, reinterpret_cast<const float4*>(ws + W.x + r0 * Hd), reinterpret_cast<float4*>(ws + W.xg),
                                       nw, Tq, Hd / 4, c.PG, c.PK);
                    RSAF_CHECK_HIP(hipGetLastError());
                }
                GemmParams p = gemm_params_plain(ws + W.xg, Wt + L.posw, ws + W.y + r0 * Hd, Tq, cg, c.PK * cg, cg, (int64_t)c.PK * cg, Hd);
                p.nz = nw * c.PG; p.nz2 = c.PG;
                p.sA1 = (int64_t)c.PG * TTq * cg; p.sA2 = (int64_t)TTq * cg;
                p.sB1 = 0; p.sB2 = (int64_t)cg * c.PK * cg;
                p.sC1 = (int64_t)Tq * Hd; p.sC2 = cg;
                p.bias = Wt + L.posb; p.sBias2 = cg; p.act = ACT_GELU;
                rc = launch_gemm_f32(p, s, "w2v2_posconv_gemm");
                if (rc) return rc;
            }
        }
        // (fused attention: this LayerNorm also reports the window's largest row norm, behind the scale of layer 0's q / k / v)
        if (fused) RSAF_CHECK_HIP(hipMemsetAsync(ws + W.win_norm, 0, sizeof(unsigned) * n, s));
        rc = ln(ws + W.x, ws + W.y, Wt + L.elng, Wt + L.elnb, ws + W.x, rows, Hd, c.eps, s, nullptr, rowwin, row0, planes_at(W.xp), true, ws + W.s_x,
                nullptr, nullptr, nullptr, fused ? bits_at(W.win_norm) : nullptr);
        if (rc) return rc;
    }
    // 6. encoder layers (post-LN)
    float* x = ws + W.x;
    for (int l = 0; l < c.L; ++l) {
        const LayerOff& lo = L.layers[l];
        const int b0 = WSTAT_LAYER0 + WSTAT_PER_LAYER * l;
        // fused q,k,v projection (A = the planes the previous LayerNorm wrote beside x).  Fused attention: the output leaves as
        // the fp16 plane pair the attention kernel multiplies, under one power of two per window (the bound over its rows)
        if (fused) {
            hipLaunchKernelGGL(qkv_scale_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, bits_at(W.win_norm), rowwin, rows,
                               wstat(b0), wstat(b0 + 5), ws + W.s_qkv);
            RSAF_CHECK_HIP(hipGetLastError());
            Out o{}; o.Cp = planes_at(W.qkv); o.c_plane = rows * 3 * Hd; o.c_scale = ws + W.s_qkv; o.cs_zs = 0; o.cs_ms = 1;
            rc = gemm3(planes_at(W.xp), rows * Hd, Hd, 0, ws + W.s_x, 0, 1, planes_at(W.wp_qkv[l]), ws + W.ws_qkv[l], (int)rows, 3 * Hd, Hd, o,
                       Wt + lo.bqkv, nullptr, 1, ACT_NONE, "w2v2_gemm", true);
            if (rc) return rc;
        } else {
            Out o{}; o.Cf = ws + W.qkv;
            rc = gemm3(planes_at(W.xp), rows * Hd, Hd, 0, ws + W.s_x, 0, 1, planes_at(W.wp_qkv[l]), ws + W.ws_qkv[l], (int)rows, 3 * Hd, Hd, o,
                       Wt + lo.bqkv, nullptr, 1, ACT_NONE, "w2v2_gemm", true);
            if (rc) return rc;
        }
        if (fused) {
            // 2 x 2 T^2 hd flops per (chunk, head)
            double att_flops = 0.0;
            for (const auto& tg : R.tgroups) att_flops += 4.0 * (tg.second - tg.first) * c.NH * (double)R.T6[tg.first] * R.T6[tg.first] * hd;
            ProfScope prof("w2v2_attn_fused", s, att_flops, 0.0);
            RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 128 * 64 * 2));
            hipLaunchKernelGGL(attn_f16x3_kernel, dim3((unsigned)(n * c.NH), (unsigned)((Tt + 127) / 128)), dim3(256), 2 * 2 * 128 * 64 * 2, s,
                               planes_at(W.qkv), rows * 3 * Hd, planes_at(W.attp), rows * Hd, rows, Tw + (int64_t)6 * n, row0, c.NH, Hd, scale,
                               ws + W.s_qkv);
            RSAF_CHECK_HIP(hipGetLastError());
        } else {
        // three launches per run of equal windows (head widths other than 64: test geometries)
        for (const auto& tg : R.tgroups) {
            const int nw = tg.second - tg.first, Tq = R.T6[tg.first], Tpq = (int)pad4(Tq);
            const int64_t r0 = R.row0[tg.first];
            const float* qkvg = ws + W.qkv + r0 * 3 * Hd;
        {   // S = scale * Q K^T per (chunk, head)
            GemmParams p = gemm_params_plain(qkvg, qkvg + Hd, ws + W.S, Tq, Tq, hd, 3 * Hd, 3 * Hd, Tpq);
            p.nz = nw * c.NH; p.nz2 = c.NH;
            p.sA1 = (int64_t)Tq * 3 * Hd; p.sA2 = hd; p.sB1 = p.sA1; p.sB2 = hd;
            p.sC1 = (int64_t)c.NH * Tq * Tpq; p.sC2 = (int64_t)Tq * Tpq;
            p.alpha = scale;
            rc = launch_gemm_f32(p, s, "w2v2_attn_gemm");
            if (rc) return rc;
        }
        {
            const int64_t srows = (int64_t)nw * c.NH * Tq;
            ProfScope prof("w2v2_softmax", s, 0.0, (double)srows * Tpq * 8);
            if (Tpq <= 256)
                hipLaunchKernelGGL(softmax_kernel<true>, dim3((unsigned)((srows + 3) / 4)), dim3(256), 0, s, ws + W.S,
                                   srows, Tq, Tpq);
            else
                hipLaunchKernelGGL(softmax_kernel<false>, dim3((unsigned)((srows + 3) / 4)), dim3(256), 0, s, ws + W.S,
                                   srows, Tq, Tpq);
            RSAF_CHECK_HIP(hipGetLastError());
        }
        {   // O = P V per (chunk, head), V is [T, hd] with N contiguous
            GemmParams p = gemm_params_plain(ws + W.S, qkvg + 2 * Hd, ws + W.att + r0 * Hd, Tq, hd, Tq, Tpq, 3 * Hd, Hd);
            p.nz = nw * c.NH; p.nz2 = c.NH; p.b_kn = 1;
            p.sA1 = (int64_t)c.NH * Tq * Tpq; p.sA2 = (int64_t)Tq * Tpq;
            p.sB1 = (int64_t)Tq * 3 * Hd; p.sB2 = hd;
            p.sC1 = (int64_t)Tq * Hd; p.sC2 = hd;
            rc = launch_gemm_f32(p, s, "w2v2_attn_gemm");
            if (rc) return rc;
        }
        }
            // (the fused kernel writes the planes itself)
            if ((rc = launch_f16x2_row_scales(ws + W.att, rows, Hd, Hd, ws + W.s_att, nullptr, nullptr, s))) return rc;
            rc = launch_split_f16x2(ws + W.att, rows, Hd, Hd, ws + W.s_att, 1, planes_at(W.attp), rows * Hd, 1, s);
            if (rc) return rc;
        }
        {   // y = attn Wo^T + bo + x ; x = LN(y), with the bound behind the scale of the ffn1 output
            Out o{}; o.Cf = ws + W.y;
            rc = gemm3(planes_at(W.attp), rows * Hd, Hd, 0, fused ? ws + W.s_qkv : ws + W.s_att, 0, 1, planes_at(W.wp_o[l]), ws + W.ws_o[l], (int)rows, Hd, Hd, o,
                       Wt + lo.bo, x, 1, ACT_NONE, "w2v2_gemm", true);
            if (rc) return rc;
            rc = ln(ws + W.y, nullptr, Wt + lo.ln1g, Wt + lo.ln1b, x, rows, Hd, c.eps, s, nullptr, nullptr, nullptr, planes_at(W.xp), true, ws + W.s_x,
                    wstat(b0 + 2), wstat(b0 + 4), ws + W.s_ffn);
            if (rc) return rc;
        }
        {   // feed forward: the GELU output only exists as planes (A of the second GEMM)
            Out o1{}; o1.Cp = planes_at(W.ffnp); o1.c_plane = rows * c.I; o1.c_scale = ws + W.s_ffn; o1.cs_zs = 0; o1.cs_ms = 1; o1.cp_panel = true;
            rc = gemm3(planes_at(W.xp), rows * Hd, Hd, 0, ws + W.s_x, 0, 1, planes_at(W.wp_1[l]), ws + W.ws_1[l], (int)rows, c.I, Hd, o1,
                       Wt + lo.b1, nullptr, 1, ACT_GELU, "w2v2_gemm", true);
            if (rc) return rc;
            Out o2{}; o2.Cf = ws + W.y;
            rc = gemm3(planes_at(W.ffnp), rows * c.I, c.I, 0, ws + W.s_ffn, 0, 1, planes_at(W.wp_2[l]), ws + W.ws_2[l], (int)rows, Hd, c.I, o2,
                       Wt + lo.b2, x, 1, ACT_NONE, "w2v2_gemm", true);
            if (rc) return rc;
            const bool last = (l == c.L - 1);
            // the last LayerNorm writes frame t of window w at out_row_start[w] + t (or packed, window after window)
            if (fused && !last) RSAF_CHECK_HIP(hipMemsetAsync(ws + W.win_norm, 0, sizeof(unsigned) * n, s));
            rc = ln(ws + W.y, nullptr, Wt + lo.ln2g, Wt + lo.ln2b, last ? out : x, rows, Hd, c.eps, s,
                    last ? out_row_start : nullptr, rowwin, row0, last ? nullptr : planes_at(W.xp), true, ws + W.s_x,
                    nullptr, nullptr, nullptr, (fused && !last) ? bits_at(W.win_norm) : nullptr);
            if (rc) return rc;
        }
    }
    return RSAF_OK;
}
