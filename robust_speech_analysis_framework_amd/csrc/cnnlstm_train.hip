// One training step of the CNN-LSTM-with-attention classifier on gfx950: forward in training mode and
// the backward pass (reference: CNNLSTM.forward under model.train() + loss.backward(),
// src/models.py:64-76,161-193 and src/dl_cv_strategies.py:118-125,241-243).
//
// Layout: activations channels-last [B][T][C] float32, exactly as in the inference path.  BatchNorm uses the
// statistics of the batch (biased variance over B*T rows, zero-padded frames included), dropout masks are
// inputs (float 0 or 1/(1-p); NULL = no dropout) so that the caller owns the random stream.
//
// Dense work goes through the exact-fp32 MFMA GEMM (gemm_f32.hip):
//   forward    conv = GEMM over the channels-last sequence (taps in K), LSTM input projections;
//   data grad  conv: the same GEMM over dy with tap-flipped, transposed weights; LSTM: dgates . W_ih (B as [K][N]);
//   weight grad  dW[M][taps*N] = sum_rows dy[row][M]^T x[row+tap][N]: both operands are transposed into [.][rows]
//              images (tap shift and sequence boundaries applied while transposing), then an NT GEMM split
//              along rows into `S` partial products that a second kernel sums in a fixed order (deterministic).
// The two recurrences are persistent kernels: the forward one of cnnlstm.hip (saving gates and cell states) and
// lstm_bwd_kernel below (W_hh register-resident as MFMA B fragments with k = gate row, dgates staged in LDS).
#include <algorithm>

#include "cnnlstm_kernels.h"
#include "gemm_f32.h"

namespace rsaf {
namespace cnntrain {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct Dims {
    int D, C, H, NC, L, act;
};

static inline int64_t pad4(int64_t n) { return (n + 3) & ~int64_t(3); }
static inline int64_t pad32(int64_t n) { return (n + 31) & ~int64_t(31); }

// ---- parameter / gradient blob ---------------------------------------------------------------------------
struct ConvP {
    int64_t w, b, g, be;
};
struct PLayout {
    ConvP c1, sc, c2, c3, c4;
    int64_t wih[4], bsum[4], whh[4];
    int64_t watt, batt, wfc, bfc, total;
};

static PLayout make_playout(const Dims& d) {
    PLayout L{};
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t s = o; o += pad4(n); return s; };
    auto conv = [&](int taps, int cin) { ConvP c; c.w = take((int64_t)d.C * taps * cin); c.b = take(d.C); c.g = take(d.C); c.be = take(d.C); return c; };
    L.c1 = conv(3, d.D);
    if (d.D != d.C) L.sc = conv(1, d.D); else L.sc = ConvP{-1, -1, -1, -1};
    L.c2 = conv(3, d.C); L.c3 = conv(3, d.C); L.c4 = conv(3, d.C);
    for (int l = 0; l < d.L; ++l) {
        const int in = l == 0 ? d.C : 2 * d.H;
        L.wih[l] = take((int64_t)8 * d.H * in);
        L.bsum[l] = take(8 * d.H);
        L.whh[l] = take((int64_t)2 * 4 * d.H * d.H);
    }
    L.watt = take(2 * d.H); L.batt = take(1);
    L.wfc = take((int64_t)d.NC * 2 * d.H); L.bfc = take(d.NC);
    L.total = o;
    return L;
}

// ---- saved activations -----------------------------------------------------------------------------------
struct SLayout {
    int64_t stat;                 // [5][3][C]: mean, biased var, rstd of bn1, shortcut bn, bn2, (block 2) bn1, bn2
    int64_t y1, ysc, a1d, y2, z1, p, y3, a3d, y4, z2, r2;
    int64_t gates[4], cst[4], hout[4], hdrop[4];
    int64_t prob, ctx, total;
};

static SLayout make_slayout(const Dims& d, int B, int T) {
    SLayout S{};
    const int64_t Tp = T / 2, n1 = pad4((int64_t)B * T * d.C), n2 = pad4((int64_t)B * Tp * d.C);
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t s = o; o += pad4(n); return s; };
    S.stat = take(5 * 3 * d.C);
    S.y1 = take(n1); S.ysc = d.D != d.C ? take(n1) : -1; S.a1d = take(n1); S.y2 = take(n1); S.z1 = take(n1);
    S.p = take(n2); S.y3 = take(n2); S.a3d = take(n2); S.y4 = take(n2); S.z2 = take(n2); S.r2 = take(n2);
    for (int l = 0; l < d.L; ++l) {
        S.gates[l] = take((int64_t)B * Tp * 8 * d.H);
        S.cst[l] = take((int64_t)B * Tp * 2 * d.H);
        S.hout[l] = take((int64_t)B * Tp * 2 * d.H);
        S.hdrop[l] = l < d.L - 1 ? take((int64_t)B * Tp * 2 * d.H) : -1;
    }
    S.prob = take((int64_t)B * Tp); S.ctx = take((int64_t)B * 2 * d.H);
    S.total = o;
    return S;
}

// split of the row dimension of a weight-gradient GEMM
struct Split { int64_t kc, s, kp; };
static Split make_split(int64_t rows) {
    int64_t kc = 1024;
    while ((rows + kc - 1) / kc > 4096) kc *= 2;
    Split sp; sp.kc = kc; sp.s = std::max<int64_t>((rows + kc - 1) / kc, 1); sp.kp = sp.s * kc;
    return sp;
}

// scratch (floats) shared by forward and backward
struct WLayout {
    int64_t bufA, bufB, bufC;           // [B*T][max(C, 2H)] activations / gradients
    int64_t t1, t2;                     // transposed images [Mmax][kp], [Nmax][kp]
    int64_t part;                       // split-K partial products
    int64_t red;                        // column-reduction partials
    int64_t wflip;                      // tap-flipped transposed conv weights [C][3][C]
    int64_t small;                      // dctx, dp, per-row partials
    int64_t total;
};

static const int RED_PARTS = 256;

static WLayout make_wlayout(const Dims& d, int B, int T) {
    WLayout W{};
    const int64_t Tp = T / 2;
    const int64_t wide = std::max<int64_t>(d.C, 2 * d.H);
    const int64_t rows = (int64_t)B * T;
    const Split sp = make_split(rows);
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t s = o; o += pad4(n); return s; };
    const int64_t nb = std::max<int64_t>((int64_t)B * T * d.C, (int64_t)B * Tp * wide);
    W.bufA = take(nb); W.bufB = take(nb); W.bufC = take(nb);
    const int64_t Mmax = std::max<int64_t>(d.C, 8 * d.H);
    const int64_t Nmax = std::max<int64_t>(std::max<int64_t>(3 * d.D, 3 * d.C), 2 * d.H);
    W.t1 = take(Mmax * sp.kp); W.t2 = take(Nmax * sp.kp);
    int64_t mn = std::max<int64_t>((int64_t)d.C * 3 * std::max(d.D, d.C), (int64_t)8 * d.H * std::max(d.C, 2 * d.H));
    W.part = take(sp.s * mn);
    W.red = take((int64_t)RED_PARTS * 2 * std::max<int64_t>(d.C, 8 * d.H) * 2);     // doubles
    W.wflip = take((int64_t)d.C * 3 * d.C);
    W.small = take(2112 + (int64_t)B * (4 * d.H + 8) + (int64_t)B * Tp + 64);   // BN sums [2][<=1024] | dctx | dwatt, dbatt partials | dp
    W.total = o;
    return W;
}

static int check_dims(const Dims& d) {
    RSAF_CHECK_ARG(d.D > 0 && d.D % 4 == 0, "input_dim must be a positive multiple of 4");
    RSAF_CHECK_ARG(d.C > 0 && d.C % 4 == 0 && d.C <= 1024, "cnn_out_channels must be a multiple of 4 in [4, 1024]");
    RSAF_CHECK_ARG(d.H == 64 || d.H == 128, "lstm_hidden_dim must be 64 or 128 (reference search space)");
    RSAF_CHECK_ARG(d.NC >= 1 && d.NC <= 16, "num_classes must be in [1, 16]");
    RSAF_CHECK_ARG(d.L >= 1 && d.L <= 4, "lstm_layers must be in [1, 4]");
    RSAF_CHECK_ARG(d.act == ACT_GELU || d.act == ACT_SILU, "activation must be gelu (1) or silu (2)");
    return RSAF_OK;
}

// ---- activation and its derivative ----------------------------------------------------------------------------
__device__ __forceinline__ float act_f(float v, int act) {
    if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v / (1.0f + expf(-v));
}
__device__ __forceinline__ float act_df(float v, int act) {
    if (act == ACT_GELU)
        return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * expf(-0.5f * v * v);
    const float s = 1.0f / (1.0f + expf(-v));
    return s * (1.0f + v * (1.0f - s));
}

// ---- per-channel reductions over rows: partial sums in double, fixed partition => deterministic ------------------
// MODE 0: (sum y, sum y^2)            -> batch statistics
// MODE 1: (sum dz, sum dz*xhat)       -> BatchNorm backward; dz = dout (PRE = 0) or dout*mask*act'(g*xhat+be) (PRE = 1)
// MODE 2: (sum a, -)                  -> bias gradients / column sums
template <int MODE, int PRE>
__global__ __launch_bounds__(256) void colred_partial_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ y,
                                                             const float* __restrict__ mask, const float* __restrict__ stat,
                                                             const float* __restrict__ g, const float* __restrict__ be, int act,
                                                             int64_t rows, int C, double* __restrict__ partial) {
    // thread = (row lane ry, channel cx); block covers 64 channels x 4 row lanes
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx;
    const int nparts = gridDim.x;
    const int64_t per = (rows + nparts - 1) / nparts;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(rows, r0 + per);
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        float mean = 0.f, rstd = 0.f, gg = 0.f, bb = 0.f;
        if (MODE == 1) { mean = stat[c]; rstd = stat[2 * C + c]; gg = g[c]; bb = be[c]; }
#pragma unroll 4
        for (int64_t r = r0 + ry; r < r1; r += 4) {
            if (MODE == 0) {
                const double v = a[r * lda + c];
                s0 += v; s1 += v * v;
            } else if (MODE == 1) {
                const float xh = (y[r * C + c] - mean) * rstd;
                float dz = a[r * lda + c];
                if (PRE) {
                    if (mask) dz *= mask[r * C + c];
                    dz *= act_df(gg * xh + bb, act);
                }
                s0 += dz; s1 += (double)dz * xh;
            } else {
                s0 += a[r * lda + c];
            }
        }
    }
    __shared__ double sh[2][4][64];
    sh[0][ry][cx] = s0; sh[1][ry][cx] = s1;
    __syncthreads();
    if (ry == 0 && c < C) {
        partial[((int64_t)blockIdx.x * 2 + 0) * C + c] = sh[0][0][cx] + sh[0][1][cx] + sh[0][2][cx] + sh[0][3][cx];
        partial[((int64_t)blockIdx.x * 2 + 1) * C + c] = sh[1][0][cx] + sh[1][1][cx] + sh[1][2][cx] + sh[1][3][cx];
    }
}

// FIN 0: stat[c] = mean, stat[C+c] = biased var, stat[2C+c] = rstd        (out0 = stat)
// FIN 1: out0[c] = sum dz*xhat (dgamma), out1[c] = sum dz (dbeta); sums[c] = both / rows for the apply kernel
// FIN 2: out0[c] = sum
template <int FIN>
__global__ __launch_bounds__(256) void colred_final_kernel(const double* __restrict__ partial, int nparts, int64_t rows, int C,
                                                           float eps, float* __restrict__ out0, float* __restrict__ out1,
                                                           float* __restrict__ sums) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int p = 0; p < nparts; ++p) { s0 += partial[((int64_t)p * 2 + 0) * C + c]; s1 += partial[((int64_t)p * 2 + 1) * C + c]; }
    if (FIN == 0) {
        const double mean = s0 / rows;
        double var = s1 / rows - mean * mean;
        if (var < 0) var = 0;
        out0[c] = (float)mean; out0[C + c] = (float)var; out0[2 * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    } else if (FIN == 1) {
        out0[c] = (float)s1; out1[c] = (float)s0;
        sums[c] = (float)(s0 / rows); sums[C + c] = (float)(s1 / rows);
    } else {
        out0[c] = (float)s0;
    }
}

// ---- elementwise kernels (float4 over channels-last rows) --------------------------------------------------------
// out = act(bn(y)) * mask
__global__ __launch_bounds__(256) void bn_act_mask_kernel(const float4* __restrict__ y, const float* __restrict__ stat,
                                                          const float* __restrict__ g, const float* __restrict__ be,
                                                          const float4* __restrict__ mask, float4* __restrict__ out, int act,
                                                          int64_t n4, int C) {
    const int C4 = C / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const float4 v = y[i];
        const float4 m = mask ? mask[i] : make_float4(1.f, 1.f, 1.f, 1.f);
        float4 o;
        o.x = act_f((v.x - stat[c + 0]) * stat[2 * C + c + 0] * g[c + 0] + be[c + 0], act) * m.x;
        o.y = act_f((v.y - stat[c + 1]) * stat[2 * C + c + 1] * g[c + 1] + be[c + 1], act) * m.y;
        o.z = act_f((v.z - stat[c + 2]) * stat[2 * C + c + 2] * g[c + 2] + be[c + 2], act) * m.z;
        o.w = act_f((v.w - stat[c + 3]) * stat[2 * C + c + 3] * g[c + 3] + be[c + 3], act) * m.w;
        out[i] = o;
    }
}

// z = bn(y) + (bn_sc(ysc) | ident);  r = act(z)
__global__ __launch_bounds__(256) void bn_add_act_kernel(const float* __restrict__ y, const float* __restrict__ stat,
                                                         const float* __restrict__ g, const float* __restrict__ be,
                                                         const float* __restrict__ ysc, const float* __restrict__ statsc,
                                                         const float* __restrict__ gsc, const float* __restrict__ besc,
                                                         const float* __restrict__ ident, int64_t ld_ident,
                                                         float* __restrict__ z, float* __restrict__ r, int act, int64_t n, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float v = (y[i] - stat[c]) * stat[2 * C + c] * g[c] + be[c];
        if (ysc) v += (ysc[i] - statsc[c]) * statsc[2 * C + c] * gsc[c] + besc[c];
        else v += ident[(i / C) * ld_ident + c];
        z[i] = v;
        r[i] = act_f(v, act);
    }
}

// dz = dr * act'(z) (+ add)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dr, const float* __restrict__ z,
                                                      float* __restrict__ dz, int act, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dz[i] = dr[i] * act_df(z[i], act);
}

// max_pool1d(2) backward fused with the activation backward of the block in front of it:
// dz1[b][2t+k][c] = (k is the arg-max of act(z1) over the pair, first wins) ? dp[b][t][c] * act'(z1) : 0; an odd last frame gets 0
__global__ __launch_bounds__(256) void pool_bwd_act_kernel(const float* __restrict__ dp, const float* __restrict__ z1,
                                                           float* __restrict__ dz1, int act, int B, int T, int Tp, int C) {
    const int64_t n = (int64_t)B * T * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t bt = i / C;
        const int t = (int)(bt % T);
        const int64_t b = bt / T;
        const int tp = t >> 1;
        float out = 0.f;
        if (tp < Tp) {
            const int64_t base = (b * T + 2 * tp) * C + c;
            const float za = z1[base], zb = z1[base + C];
            const float ra = act_f(za, act), rb = act_f(zb, act);
            const bool second = rb > ra;                  // first index wins ties (and NaN never wins): ATen max_pool1d
            if (second == ((t & 1) != 0)) out = dp[(b * Tp + tp) * C + c] * act_df((t & 1) ? zb : za, act);
        }
        dz1[i] = out;
    }
}

// BatchNorm backward, second half: dy = g*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)); dz as in colred MODE 1
template <int PRE>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ mask,
                                                           const float* __restrict__ y, const float* __restrict__ stat,
                                                           const float* __restrict__ g, const float* __restrict__ be,
                                                           const float* __restrict__ sums, float* __restrict__ dy, int act,
                                                           int64_t n, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float rstd = stat[2 * C + c];
        const float xh = (y[i] - stat[c]) * rstd;
        float dz = dout[i];
        if (PRE) {
            if (mask) dz *= mask[i];
            dz *= act_df(g[c] * xh + be[c], act);
        }
        dy[i] = g[c] * rstd * (dz - sums[c] - xh * sums[C + c]);
    }
}

__global__ __launch_bounds__(256) void mul_kernel(const float4* __restrict__ a, const float4* __restrict__ m,
                                                  float4* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = a[i], k = m[i];
        out[i] = make_float4(x.x * k.x, x.y * k.y, x.z * k.z, x.w * k.w);
    }
}

__global__ __launch_bounds__(256) void add_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                  float4* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = a[i], k = b[i];
        out[i] = make_float4(x.x + k.x, x.y + k.y, x.z + k.z, x.w + k.w);
    }
}

// max_pool1d(2), channels-last
__global__ __launch_bounds__(256) void pool2_kernel(const float4* __restrict__ x, float4* __restrict__ y, int B, int T, int Tp, int C4) {
    const int64_t n = (int64_t)B * Tp * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const int64_t bt = i / C4;
        const int t = (int)(bt % Tp);
        const int64_t b = bt / Tp;
        const float4 a = x[(b * T + 2 * t) * C4 + c];
        const float4 d = x[(b * T + 2 * t + 1) * C4 + c];
        y[i] = make_float4(fmaxf(a.x, d.x), fmaxf(a.y, d.y), fmaxf(a.z, d.z), fmaxf(a.w, d.w));
    }
}

// w [Cout][3][Cin] -> wf [Cin][3][Cout], wf[ci][j][co] = w[co][2-j][ci]  (data gradient of a k=3/pad=1 convolution)
__global__ __launch_bounds__(256) void flip_taps_kernel(const float* __restrict__ w, float* __restrict__ wf, int Cout, int Cin) {
    const int64_t n = (int64_t)Cout * 3 * Cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % Cout);
        const int j = (int)((i / Cout) % 3);
        const int ci = (int)(i / (3 * (int64_t)Cout));
        wf[i] = w[((int64_t)co * 3 + (2 - j)) * Cin + ci];
    }
}

// dst[c][b*T + t] = src[(b*T + t + shift)*ld + c] if 0 <= t + shift < T else 0; columns [B*T, kp) are zero
__global__ __launch_bounds__(256) void transpose_shift_kernel(const float* __restrict__ src, int64_t ld, int ncols, int B, int T,
                                                              int shift, float* __restrict__ dst, int64_t kp) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int64_t k0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int64_t rows = (int64_t)B * T;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t k = k0 + ty + 8 * i;
        const int c = c0 + tx;
        float v = 0.f;
        if (k < rows && c < ncols) {
            const int t = (int)(k % T) + shift;
            if (t >= 0 && t < T) v = src[(k + shift) * ld + c];
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i;
        const int64_t k = k0 + tx;
        if (c < ncols && k < kp) dst[(int64_t)c * kp + k] = tile[tx][ty + 8 * i];
    }
}

__global__ __launch_bounds__(256) void sum_splits_kernel(const float* __restrict__ part, int S, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += part[(int64_t)z * n + i];
        out[i] = s;
    }
}

// ---- attention pooling, training forward: probabilities and pooled context are kept -----------------------------------
constexpr int ATT_WAVES = 16;     // one 1024-thread workgroup per sequence: 16 row streams in flight on its CU

template <int NF>   // 2H = 64*NF
__global__ __launch_bounds__(ATT_WAVES * 64) void attnpool_train_kernel(const float* __restrict__ seq, const float* __restrict__ watt,
                                                                        const float* __restrict__ batt, float* __restrict__ prob,
                                                                        float* __restrict__ ctx_out, int T) {
    constexpr int F = 64 * NF;
    __shared__ float s_m[ATT_WAVES], s_l[ATT_WAVES], s_ctx[ATT_WAVES][F];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* sb = seq + (int64_t)b * T * F;
    float* pb = prob + (int64_t)b * T;
    float wa[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) wa[i] = watt[lane + 64 * i];
    const float ba = batt[0];
    float m = -INFINITY, l = 0.f;
#pragma unroll 2
    for (int t = w; t < T; t += ATT_WAVES) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NF; ++i) d += sb[(int64_t)t * F + lane + 64 * i] * wa[i];
        const float sc = wave_sum(d) + ba;
        if (lane == 0) pb[t] = sc;
        const float mn = fmaxf(m, sc);
        l = l * expf(m - mn) + expf(sc - mn);
        m = mn;
    }
    if (lane == 0) { s_m[w] = m; s_l[w] = l; }
    __syncthreads();
    float M = s_m[0];
#pragma unroll
    for (int k = 1; k < ATT_WAVES; ++k) M = fmaxf(M, s_m[k]);
    float Lt = 0.f;
#pragma unroll
    for (int k = 0; k < ATT_WAVES; ++k) Lt += s_l[k] > 0.f ? s_l[k] * expf(s_m[k] - M) : 0.f;
    float ctx[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) ctx[i] = 0.f;
#pragma unroll 2
    for (int t = w; t < T; t += ATT_WAVES) {
        // lane 0 of this wave wrote pb[t] in the first pass: read it past the (non-coherent) vector L1
        const float p = expf(__hip_atomic_load(pb + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - M) / Lt;
#pragma unroll
        for (int i = 0; i < NF; ++i) ctx[i] += p * sb[(int64_t)t * F + lane + 64 * i];
        if (lane == 0) pb[t] = p;                    // after this wave's own read of the score (program order, same address)
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) s_ctx[w][lane + 64 * i] = ctx[i];
    __syncthreads();
    const int f = threadIdx.x;
    if (f < F) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < ATT_WAVES; ++k) a += s_ctx[k][f];
        ctx_out[(int64_t)b * F + f] = a;
    }
}

// logits = (ctx * mask) . wfc^T + bfc
__global__ __launch_bounds__(256) void fc_fwd_kernel(const float* __restrict__ ctx, const float* __restrict__ mask,
                                                     const float* __restrict__ wfc, const float* __restrict__ bfc,
                                                     float* __restrict__ logits, int F, int NC) {
    __shared__ float s_red[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = 0; c < NC; ++c) {
        float part = 0.f;
        for (int f = threadIdx.x; f < F; f += 256) {
            float v = ctx[(int64_t)b * F + f];
            if (mask) v *= mask[(int64_t)b * F + f];
            part += v * wfc[(int64_t)c * F + f];
        }
        part = wave_sum(part);
        __syncthreads();
        if (lane == 0) s_red[w] = part;
        __syncthreads();
        if (threadIdx.x == 0) logits[(int64_t)b * NC + c] = s_red[0] + s_red[1] + s_red[2] + s_red[3] + bfc[c];
    }
}

// classifier backward: thread f owns feature f.  dwfc[c][f], dbfc[c], dctx[b][f]
__global__ __launch_bounds__(256) void fc_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ ctx,
                                                     const float* __restrict__ mask, const float* __restrict__ wfc,
                                                     float* __restrict__ dwfc, float* __restrict__ dbfc, float* __restrict__ dctx,
                                                     int B, int F, int NC) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < F) {
        for (int c = 0; c < NC; ++c) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) {
                float v = ctx[(int64_t)b * F + f];
                if (mask) v *= mask[(int64_t)b * F + f];
                s += dlogits[(int64_t)b * NC + c] * v;
            }
            dwfc[(int64_t)c * F + f] = s;
        }
        for (int b = 0; b < B; ++b) {
            float s = 0.f;
            for (int c = 0; c < NC; ++c) s += dlogits[(int64_t)b * NC + c] * wfc[(int64_t)c * F + f];
            if (mask) s *= mask[(int64_t)b * F + f];
            dctx[(int64_t)b * F + f] = s;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < NC) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dlogits[(int64_t)b * NC + threadIdx.x];
        dbfc[threadIdx.x] = s;
    }
}

// attention pooling backward for one sequence: dh[t][f] = p_t*dctx[f] + ds_t*wa[f], ds_t = p_t*(dp_t - sum_t' p_t' dp_t'),
// dp_t = dctx . h_t; per-sequence partials of dwatt[f] = sum_t ds_t h_t[f] and dbatt = sum_t ds_t
template <int NF>
__global__ __launch_bounds__(ATT_WAVES * 64) void attn_bwd_kernel(const float* __restrict__ seq, const float* __restrict__ prob,
                                                                  const float* __restrict__ dctx, const float* __restrict__ watt,
                                                                  float* __restrict__ dp_scratch, float* __restrict__ dseq,
                                                                  float* __restrict__ dwatt_part, float* __restrict__ dbatt_part, int T) {
    constexpr int F = 64 * NF;
    __shared__ float s_dot[ATT_WAVES], s_db[ATT_WAVES], s_dw[ATT_WAVES][F];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* sb = seq + (int64_t)b * T * F;
    const float* pb = prob + (int64_t)b * T;
    float* dpb = dp_scratch + (int64_t)b * T;
    float* db = dseq + (int64_t)b * T * F;
    float dc[NF], wa[NF], dw[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) { dc[i] = dctx[(int64_t)b * F + lane + 64 * i]; wa[i] = watt[lane + 64 * i]; dw[i] = 0.f; }
    float dot = 0.f;
#pragma unroll 2
    for (int t = w; t < T; t += ATT_WAVES) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NF; ++i) d += sb[(int64_t)t * F + lane + 64 * i] * dc[i];
        d = wave_sum(d);
        if (lane == 0) dpb[t] = d;
        dot += pb[t] * d;
    }
    if (lane == 0) s_dot[w] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int k = 0; k < ATT_WAVES; ++k) dot += s_dot[k];
    float dbs = 0.f;
#pragma unroll 2
    for (int t = w; t < T; t += ATT_WAVES) {
        const float p = pb[t];
        const float ds = p * (__hip_atomic_load(dpb + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - dot);   // written by lane 0 above
        dbs += ds;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const float h = sb[(int64_t)t * F + lane + 64 * i];
            db[(int64_t)t * F + lane + 64 * i] = p * dc[i] + ds * wa[i];
            dw[i] += ds * h;
        }
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) s_dw[w][lane + 64 * i] = dw[i];
    if (lane == 0) s_db[w] = dbs;
    __syncthreads();
    const int f = threadIdx.x;
    if (f < F) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < ATT_WAVES; ++k) a += s_dw[k][f];
        dwatt_part[(int64_t)b * F + f] = a;
    }
    if (f == 0) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < ATT_WAVES; ++k) a += s_db[k];
        dbatt_part[b] = a;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- persistent LSTM backward recurrence (BPTT) ---------------------------------------------------------------------
// One workgroup = 16 batch rows of one direction; wave w owns hidden units 16w..16w+15.  Walks the time steps in the
// reverse of the forward order.  Per step, lane-local: dh = dh_out + dh_rec, gate/cell derivatives from the saved
// post-activation gates and cell states; the pre-activation gate gradients overwrite the saved gates (they are the
// operand of the weight / input gradients afterwards) and go to LDS as the MFMA A operand of
//     dh_rec[row][unit] = sum_k dgates[row][k] * W_hh[k][unit],   k over the 4H gate rows,
// with W_hh register-resident as B fragments (H registers per lane).
template <int H>
__global__ __launch_bounds__(H / 16 * 64) void lstm_bwd_kernel(float* gates, const float* __restrict__ cst,
                                                               const float* __restrict__ dh_out, const float* __restrict__ whh,
                                                               int B, int T) {
    constexpr int LDG = 4 * H + 4;
    constexpr int KG = 4 * H / 16;                  // k-groups of 16 gate rows
    extern __shared__ __attribute__((aligned(16))) float dgbuf[];     // [2][16][LDG]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, col = lane & 15, q = lane >> 4;
    const int dir = blockIdx.y, b0 = blockIdx.x * 16, unit = 16 * w + col;

    // B fragments: breg[4g + j] = W_hh[dir][16g + 4q + j][unit]
    float breg[KG * 4];
    const float* wd = whh + (int64_t)dir * 4 * H * H;
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) breg[4 * g + j] = wd[(int64_t)(16 * g + 4 * q + j) * H + unit];
    for (int i = tid; i < 2 * 16 * LDG; i += H / 16 * 64) dgbuf[i] = 0.0f;

    int brow[4];
    int64_t goff[4], hoff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + q * 4 + r;
        brow[r] = b < B;
        const int bc = b < B ? b : B - 1;
        goff[r] = (int64_t)bc * T * 8 * H + dir * 4 * H + unit;
        hoff[r] = (int64_t)bc * T * 2 * H + dir * H + unit;
    }
    float dcc[4] = {0.f, 0.f, 0.f, 0.f};            // dL/dc carried to the previous forward step
    f32x4 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : T - 1 - s;           // reverse of the forward order
        const int tprev = dir ? t + 1 : t - 1;       // the step the forward pass ran just before t
        const bool first = dir ? (t == T - 1) : (t == 0);
        float* dgw = dgbuf + (cur ^ 1) * 16 * LDG;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* gp = gates + goff[r] + (int64_t)t * 8 * H;
            const float ig = gp[0], fg = gp[H], gg = gp[2 * H], og = gp[3 * H];
            const float c = cst[hoff[r] + (int64_t)t * 2 * H];
            const float cpl = cst[hoff[r] + (int64_t)(first ? t : tprev) * 2 * H];
            const float cp = first ? 0.f : cpl;
            const float dh = dh_out[hoff[r] + (int64_t)t * 2 * H] + (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
            const float tc = tanhf(c);
            const float dc = dcc[r] + dh * og * (1.0f - tc * tc);
            const float d_o = dh * tc * og * (1.0f - og);
            const float d_i = dc * gg * ig * (1.0f - ig);
            const float d_f = dc * cp * fg * (1.0f - fg);
            const float d_g = dc * ig * (1.0f - gg * gg);
            dcc[r] = dc * fg;
            const int row = q * 4 + r;
            dgw[row * LDG + unit] = d_i; dgw[row * LDG + H + unit] = d_f;
            dgw[row * LDG + 2 * H + unit] = d_g; dgw[row * LDG + 3 * H + unit] = d_o;
            // rows past B are clamped copies of row B-1: they must NOT store here, because this loop reads the gates of
            // step t row by row and a copy in an earlier register row would overwrite them before the real row reads
            if (brow[r]) {
                float* go = gates + goff[r] + (int64_t)t * 8 * H;
                go[0] = d_i; go[H] = d_f; go[2 * H] = d_g; go[3 * H] = d_o;
            }
        }
        lds_barrier();
        cur ^= 1;
        const float* dgr = dgbuf + cur * 16 * LDG;
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const float4 a4 = *reinterpret_cast<const float4*>(&dgr[col * LDG + 16 * g + 4 * q]);
            acc[g & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, breg[4 * g + 0], acc[g & 3], 0, 0, 0);
            acc[g & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, breg[4 * g + 1], acc[g & 3], 0, 0, 0);
            acc[g & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, breg[4 * g + 2], acc[g & 3], 0, 0, 0);
            acc[g & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, breg[4 * g + 3], acc[g & 3], 0, 0, 0);
        }
    }
}

// ---- the same backward recurrence for small batches: 4 batch rows per workgroup ---------------------------------------
// Wave w owns hidden units 16w..16w+15; lane (q, u) = (lane / 16, lane % 16) owns cell (row q, unit 16w + u): lane-local
// gate / cell derivatives, dgates to HBM (in place) and to LDS (double-buffered: one LDS-only barrier per step).
// dh_rec = dgates . W_hh with v_mfma_f32_4x4x1_16B_f32 (row i of every block = batch row i): the 16 blocks of an
// instruction are 4 gates x 4 groups of 4 units, i.e. lane (g, u) contracts the H rows of gate g against unit u (its
// W_hh column slice is register-resident, H registers) and a two-step butterfly over the gate lanes completes the sum
// in a fixed order.  No partial products through LDS.
template <int H>
__global__ __launch_bounds__(H / 16 * 64) void lstm_bwd4_kernel(float* gates, const float* __restrict__ cst,
                                                                const float* __restrict__ dh_out, const float* __restrict__ whh,
                                                                int B, int T) {
    constexpr int NW = H / 16;
    constexpr int LDG = 4 * H + 20;                 // row stride = 20 banks (mod 32): the four rows of a fragment read and the
                                                    // 16-lane row groups of the cell owners' writes land on distinct banks
    __shared__ __attribute__((aligned(16))) float dg[2][4][LDG];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, u = lane & 15;
    const int dir = blockIdx.y, b0 = blockIdx.x * 4;
    const int unit = 16 * w + u;
    const int arow = lane & 3;

    // breg[m] = W_hh[dir][q*H + m][unit]   (q as the gate of this lane's blocks)
    float breg[H];
    {
        const float* wd = whh + (int64_t)dir * 4 * H * H + (int64_t)(q * H) * H + unit;
#pragma unroll
        for (int m = 0; m < H; ++m) breg[m] = wd[(int64_t)m * H];
    }
    const int bo = min(b0 + q, B - 1);              // rows past B replicate row B-1 bit for bit (see lstm_rec_kernel)
    const int64_t goff = (int64_t)bo * T * 8 * H + dir * 4 * H + unit;
    const int64_t hoff = (int64_t)bo * T * 2 * H + dir * H + unit;
    float dcc = 0.f, dh_rec = 0.f;
    float ig, fg, gg, og, cc, dho;                  // operands of the current step (prefetched)
    {
        const int t = dir ? 0 : T - 1;
        const float* gp = gates + goff + (int64_t)t * 8 * H;
        ig = gp[0]; fg = gp[H]; gg = gp[2 * H]; og = gp[3 * H];
        cc = cst[hoff + (int64_t)t * 2 * H];
        dho = dh_out[hoff + (int64_t)t * 2 * H];
    }
    vmem_drain();
    int cur = 0;
    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : T - 1 - s;           // reverse of the forward order
        const int tprev = dir ? t + 1 : t - 1;       // the step the forward pass ran just before t == the next step here
        const bool last = s + 1 == T;
        // next step's operands (branch-free: the last step reloads its own and zeroes c_prev)
        const int tl = last ? t : tprev;
        const float* gp = gates + goff + (int64_t)tl * 8 * H;
        const float nig = gp[0], nfg = gp[H], ngg = gp[2 * H], nog = gp[3 * H];
        const float ncl = cst[hoff + (int64_t)tl * 2 * H];
        const float ncc = last ? 0.f : ncl;
        const float ndho = dh_out[hoff + (int64_t)tl * 2 * H];

        const float dh = dho + dh_rec;
        const float tc = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * cc)) - 1.0f;
        const float dc = dcc + dh * og * (1.0f - tc * tc);
        const float d_o = dh * tc * og * (1.0f - og);
        const float d_i = dc * gg * ig * (1.0f - ig);
        const float d_f = dc * ncc * fg * (1.0f - fg);           // c of the previous forward step (0 at the first)
        const float d_g = dc * ig * (1.0f - gg * gg);
        dcc = dc * fg;
        float* dgw = &dg[cur][q][0];
        dgw[unit] = d_i; dgw[H + unit] = d_f; dgw[2 * H + unit] = d_g; dgw[3 * H + unit] = d_o;
        {
            float* go = gates + goff + (int64_t)t * 8 * H;
            go[0] = d_i; go[H] = d_f; go[2 * H] = d_g; go[3 * H] = d_o;
        }
        lds_barrier();
        // A operand: lane 4*blk + i carries batch row i; the block's gate is this lane's q
        const float* dgr = &dg[cur][arow][q * H];
        f32x4 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 ab[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ab[0][j] = *reinterpret_cast<const float4*>(dgr + 4 * j);
#pragma unroll
        for (int kb = 0; kb < H / 32; ++kb) {
            if (kb + 1 < H / 32) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ab[(kb + 1) & 1][j] = *reinterpret_cast<const float4*>(dgr + 32 * (kb + 1) + 4 * j);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 a4 = ab[kb & 1][j];
                const int m = 32 * kb + 4 * j;
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.x, breg[m + 0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.y, breg[m + 1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.z, breg[m + 2], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.w, breg[m + 3], acc[3], 0, 0, 0);
            }
        }
        // lane (g, u), register r: sum over the rows of gate g for (batch row r, unit u).  Butterfly over g; lane (q, u)
        // keeps batch row q.
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            pr[r] = v;
        }
        dh_rec = q == 0 ? pr[0] : q == 1 ? pr[1] : q == 2 ? pr[2] : pr[3];
        cur ^= 1;
        ig = nig; fg = nfg; gg = ngg; og = nog; cc = ncc; dho = ndho;
    }
}

// ---- host-side helpers ------------------------------------------------------------------------------------------
static inline int ew_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 32)); }

static int conv3(const float* x, const float* wk, const float* bias, float* y, int B, int T, int Cin, int Cout, hipStream_t s,
                 const char* tag) {
    GemmParams p = gemm_params_plain(x - Cin, wk, y, T, Cout, 3 * Cin, Cin, 3 * Cin, Cout);
    p.bias = bias;
    p.nz = B; p.nz2 = 1; p.sA1 = (int64_t)T * Cin; p.sC1 = (int64_t)T * Cout;
    p.a_pad_k = Cin;
    return launch_gemm_f32(p, s, tag);
}

struct Ctx {
    Dims d;
    hipStream_t s;
    float* ws;
    WLayout W;
};

static int bn_stats(const Ctx& c, const float* y, int64_t rows, float* stat) {
    const int C = c.d.C;
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(RED_PARTS, rows / 128));
    double* partial = reinterpret_cast<double*>(c.ws + c.W.red);
    ProfScope prof("train_bn_reduce", c.s, 0.0, (double)rows * C * 4);
    hipLaunchKernelGGL((colred_partial_kernel<0, 0>), dim3(parts, (C + 63) / 64), dim3(256), 0, c.s, y, (int64_t)C, nullptr, nullptr,
                       nullptr, nullptr, nullptr, 0, rows, C, partial);
    hipLaunchKernelGGL((colred_final_kernel<0>), dim3((C + 255) / 256), dim3(256), 0, c.s, partial, parts, rows, C, 1e-5f, stat,
                       nullptr, nullptr);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

static int colsum(const Ctx& c, const float* a, int64_t lda, int64_t rows, int N, float* out) {
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(RED_PARTS, rows / 128));
    double* partial = reinterpret_cast<double*>(c.ws + c.W.red);
    ProfScope prof("train_bn_reduce", c.s, 0.0, (double)rows * N * 4);
    hipLaunchKernelGGL((colred_partial_kernel<2, 0>), dim3(parts, (N + 63) / 64), dim3(256), 0, c.s, a, lda, nullptr, nullptr, nullptr,
                       nullptr, nullptr, 0, rows, N, partial);
    hipLaunchKernelGGL((colred_final_kernel<2>), dim3((N + 255) / 256), dim3(256), 0, c.s, partial, parts, rows, N, 0.f, out, nullptr,
                       nullptr);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

// BatchNorm backward: dgamma, dbeta and dy (dy may alias dout)
static int bn_backward(const Ctx& c, const float* dout, const float* mask, bool pre, const float* y, const float* stat,
                       const float* g, const float* be, int64_t rows, float* dgamma, float* dbeta, float* dy) {
    const int C = c.d.C;
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(RED_PARTS, rows / 128));
    double* partial = reinterpret_cast<double*>(c.ws + c.W.red);
    float* sums = c.ws + c.W.small;                      // [2][C] (C <= 1024 fits: small >= 64 + ...)
    {
        ProfScope prof("train_bn_reduce", c.s, 0.0, (double)rows * C * 8);
        if (pre)
            hipLaunchKernelGGL((colred_partial_kernel<1, 1>), dim3(parts, (C + 63) / 64), dim3(256), 0, c.s, dout, (int64_t)C, y, mask, stat,
                               g, be, c.d.act, rows, C, partial);
        else
            hipLaunchKernelGGL((colred_partial_kernel<1, 0>), dim3(parts, (C + 63) / 64), dim3(256), 0, c.s, dout, (int64_t)C, y, mask, stat,
                               g, be, c.d.act, rows, C, partial);
        hipLaunchKernelGGL((colred_final_kernel<1>), dim3((C + 255) / 256), dim3(256), 0, c.s, partial, parts, rows, C, 0.f, dgamma, dbeta,
                           sums);
    }
    {
        ProfScope prof("train_elementwise", c.s, 0.0, (double)rows * C * 12);
        const int64_t n = rows * C;
        if (pre)
            hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), dim3(ew_blocks(n)), dim3(256), 0, c.s, dout, mask, y, stat, g, be, sums, dy, c.d.act, n, C);
        else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<0>), dim3(ew_blocks(n)), dim3(256), 0, c.s, dout, mask, y, stat, g, be, sums, dy, c.d.act, n, C);
    }
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

static int transpose_shift(const Ctx& c, const float* src, int64_t ld, int ncols, int B, int T, int shift, float* dst, int64_t kp) {
    ProfScope prof("train_transpose", c.s, 0.0, (double)B * T * ncols * 8);
    dim3 grid((unsigned)(kp / 32), (ncols + 31) / 32);
    hipLaunchKernelGGL(transpose_shift_kernel, grid, dim3(256), 0, c.s, src, ld, ncols, B, T, shift, dst, kp);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

// dW[M][taps*N] (tap-major) = sum over rows of dy[row][M]^T xin[row + tap - taps/2][N], sequence boundaries respected;
// `shift0` replaces the tap shift when taps == 1 (recurrent weight gradients use -1 / +1)
static int wgrad(const Ctx& c, const float* dy, int64_t ld_dy, int M, const float* xin, int64_t ld_x, int N, int taps, int shift0,
                 int B, int T, float* out) {
    const int64_t rows = (int64_t)B * T;
    const Split sp = make_split(rows);
    float* t1 = c.ws + c.W.t1;
    float* t2 = c.ws + c.W.t2;
    float* part = c.ws + c.W.part;
    int rc = transpose_shift(c, dy, ld_dy, M, B, T, 0, t1, sp.kp);
    if (rc) return rc;
    for (int j = 0; j < taps; ++j) {
        rc = transpose_shift(c, xin, ld_x, N, B, T, taps == 1 ? shift0 : j - taps / 2, t2 + (int64_t)j * N * sp.kp, sp.kp);
        if (rc) return rc;
    }
    const int NN = taps * N;
    GemmParams p = gemm_params_plain(t1, t2, sp.s == 1 ? out : part, M, NN, (int)sp.kc, sp.kp, sp.kp, NN);
    p.nz = (int)sp.s; p.nz2 = 1; p.sA1 = sp.kc; p.sB1 = sp.kc; p.sC1 = (int64_t)M * NN;
    rc = launch_gemm_f32(p, c.s, "train_wgrad_gemm");
    if (rc) return rc;
    if (sp.s > 1) {
        const int64_t n = (int64_t)M * NN;
        ProfScope prof("train_elementwise", c.s, 0.0, (double)n * sp.s * 4);
        hipLaunchKernelGGL(sum_splits_kernel, dim3(ew_blocks(n)), dim3(256), 0, c.s, part, (int)sp.s, n, out);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    return RSAF_OK;
}

// data gradient of a k=3/pad=1 convolution: dx[B][T][Cin] from dy[B][T][Cout] and w[Cout][3][Cin]
static int conv3_dgrad(const Ctx& c, const float* dy, const float* w, float* dx, int B, int T, int Cin, int Cout) {
    float* wf = c.ws + c.W.wflip;
    {
        const int64_t n = (int64_t)Cout * 3 * Cin;
        ProfScope prof("train_elementwise", c.s, 0.0, (double)n * 8);
        hipLaunchKernelGGL(flip_taps_kernel, dim3(ew_blocks(n)), dim3(256), 0, c.s, w, wf, Cout, Cin);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    return conv3(dy, wf, nullptr, dx, B, T, Cout, Cin, c.s, "train_dgrad_gemm");
}

#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

}  // namespace cnntrain
}  // namespace rsaf

using namespace rsaf;
using namespace rsaf::cnntrain;

extern "C" {

int64_t rsaf_cnnlstm_train_param_floats(int input_dim, int channels, int hidden, int num_classes, int lstm_layers) {
    Dims d{input_dim, channels, hidden, num_classes, lstm_layers, ACT_SILU};
    if (check_dims(d) != RSAF_OK) return -1;
    return make_playout(d).total;
}

int rsaf_cnnlstm_train_param_offsets(int input_dim, int channels, int hidden, int num_classes, int lstm_layers,
                                     int64_t* offsets_host, int cap, int* n_host) {
    Dims d{input_dim, channels, hidden, num_classes, lstm_layers, ACT_SILU};
    TRY(check_dims(d));
    RSAF_CHECK_ARG(offsets_host && n_host, "NULL output");
    const PLayout L = make_playout(d);
    int64_t v[20 + 12 + 4];
    int n = 0;
    for (const ConvP* c : {&L.c1, &L.sc, &L.c2, &L.c3, &L.c4}) { v[n++] = c->w; v[n++] = c->b; v[n++] = c->g; v[n++] = c->be; }
    for (int l = 0; l < d.L; ++l) { v[n++] = L.wih[l]; v[n++] = L.bsum[l]; v[n++] = L.whh[l]; }
    v[n++] = L.watt; v[n++] = L.batt; v[n++] = L.wfc; v[n++] = L.bfc;
    RSAF_CHECK_ARG(cap >= n, "offsets_host too small");
    for (int i = 0; i < n; ++i) offsets_host[i] = v[i];
    *n_host = n;
    return RSAF_OK;
}

int64_t rsaf_cnnlstm_train_saved_floats(int B, int T, int input_dim, int channels, int hidden, int lstm_layers) {
    if (B <= 0 || T < 2) return -1;
    Dims d{input_dim, channels, hidden, 2, lstm_layers, ACT_SILU};
    if (check_dims(d) != RSAF_OK) return -1;
    return make_slayout(d, B, T).total;
}

int64_t rsaf_cnnlstm_train_scratch_floats(int B, int T, int input_dim, int channels, int hidden, int lstm_layers) {
    if (B <= 0 || T < 2) return -1;
    Dims d{input_dim, channels, hidden, 2, lstm_layers, ACT_SILU};
    if (check_dims(d) != RSAF_OK) return -1;
    return make_wlayout(d, B, T).total;
}

int rsaf_cnnlstm_train_forward(const float* x, int B, int T, int input_dim, int channels, int hidden, int num_classes,
                               int lstm_layers, int act, const float* params, const float* mask_block1,
                               const float* mask_block2, const float* const* mask_lstm_host, const float* mask_fc, float* saved,
                               int64_t saved_floats, float* scratch, int64_t scratch_floats, float* logits,
                               float* bn_stats_out, rsaf_stream_t stream) {
    Dims d{input_dim, channels, hidden, num_classes, lstm_layers, act};
    TRY(check_dims(d));
    RSAF_CHECK_ARG(B >= 1 && B <= 65535, "batch must be in [1, 65535]");
    RSAF_CHECK_ARG(T >= 2, "sequence length must be >= 2 (max_pool1d(2) of the reference needs it)");
    RSAF_CHECK_ARG((int64_t)B * T <= 0x3fffffffLL, "B*T too large");
    RSAF_CHECK_ARG(x && params && saved && scratch && logits, "NULL pointer");
    const SLayout S = make_slayout(d, B, T);
    const WLayout WL = make_wlayout(d, B, T);
    if (saved_floats < S.total || scratch_floats < WL.total) {
        set_error("rsaf_cnnlstm_train_forward: saved/scratch buffer too small");
        return RSAF_ERR_WORKSPACE;
    }
    const PLayout L = make_playout(d);
    hipStream_t s = (hipStream_t)stream;
    Ctx c{d, s, scratch, WL};
    const int D = d.D, C = d.C, H = d.H, Tp = T / 2;
    const int64_t rows = (int64_t)B * T, rows2 = (int64_t)B * Tp;
    const float* P = params;
    float* st = saved + S.stat;
    auto stat = [&](int i) { return st + (int64_t)i * 3 * C; };
    float* r1 = scratch + WL.bufA;

    // ---- res_block1 (src/models.py:64-76) --------------------------------------------------------------------
    TRY(conv3(x, P + L.c1.w, P + L.c1.b, saved + S.y1, B, T, D, C, s, "train_conv_gemm"));
    TRY(bn_stats(c, saved + S.y1, rows, stat(0)));
    {
        ProfScope prof("train_elementwise", s, 0.0, (double)rows * C * 12);
        hipLaunchKernelGGL(bn_act_mask_kernel, dim3(ew_blocks(rows * C / 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(saved + S.y1),
                           stat(0), P + L.c1.g, P + L.c1.be, reinterpret_cast<const float4*>(mask_block1),
                           reinterpret_cast<float4*>(saved + S.a1d), act, rows * C / 4, C);
    }
    if (D != C) {
        GemmParams p = gemm_params_plain(x, P + L.sc.w, saved + S.ysc, (int)rows, C, D, D, D, C);
        p.bias = P + L.sc.b;
        TRY(launch_gemm_f32(p, s, "train_conv_gemm"));
        TRY(bn_stats(c, saved + S.ysc, rows, stat(1)));
    }
    TRY(conv3(saved + S.a1d, P + L.c2.w, P + L.c2.b, saved + S.y2, B, T, C, C, s, "train_conv_gemm"));
    TRY(bn_stats(c, saved + S.y2, rows, stat(2)));
    {
        ProfScope prof("train_elementwise", s, 0.0, (double)rows * C * 16);
        hipLaunchKernelGGL(bn_add_act_kernel, dim3(ew_blocks(rows * C)), dim3(256), 0, s, saved + S.y2, stat(2), P + L.c2.g, P + L.c2.be,
                           D != C ? saved + S.ysc : nullptr, stat(1), D != C ? P + L.sc.g : nullptr, D != C ? P + L.sc.be : nullptr,
                           x, (int64_t)D, saved + S.z1, r1, act, rows * C, C);
    }
    // ---- max_pool1d(2) (:177) -----------------------------------------------------------------------------------
    {
        ProfScope prof("train_elementwise", s, 0.0, (double)rows * C * 6);
        hipLaunchKernelGGL(pool2_kernel, dim3(ew_blocks(rows2 * C / 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(r1),
                           reinterpret_cast<float4*>(saved + S.p), B, T, Tp, C / 4);
    }
    // ---- res_block2, identity shortcut (:178) ----------------------------------------------------------------------
    TRY(conv3(saved + S.p, P + L.c3.w, P + L.c3.b, saved + S.y3, B, Tp, C, C, s, "train_conv_gemm"));
    TRY(bn_stats(c, saved + S.y3, rows2, stat(3)));
    {
        ProfScope prof("train_elementwise", s, 0.0, (double)rows2 * C * 12);
        hipLaunchKernelGGL(bn_act_mask_kernel, dim3(ew_blocks(rows2 * C / 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(saved + S.y3),
                           stat(3), P + L.c3.g, P + L.c3.be, reinterpret_cast<const float4*>(mask_block2),
                           reinterpret_cast<float4*>(saved + S.a3d), act, rows2 * C / 4, C);
    }
    TRY(conv3(saved + S.a3d, P + L.c4.w, P + L.c4.b, saved + S.y4, B, Tp, C, C, s, "train_conv_gemm"));
    TRY(bn_stats(c, saved + S.y4, rows2, stat(4)));
    {
        ProfScope prof("train_elementwise", s, 0.0, (double)rows2 * C * 16);
        hipLaunchKernelGGL(bn_add_act_kernel, dim3(ew_blocks(rows2 * C)), dim3(256), 0, s, saved + S.y4, stat(4), P + L.c4.g, P + L.c4.be,
                           nullptr, nullptr, nullptr, nullptr, saved + S.p, (int64_t)C, saved + S.z2, saved + S.r2, act, rows2 * C, C);
    }
    RSAF_CHECK_HIP(hipGetLastError());
    // ---- LSTM (:184): dropout between the layers ----------------------------------------------------------------------
    const float* lin = saved + S.r2;
    int in = C;
    for (int l = 0; l < d.L; ++l) {
        float* gates = saved + S.gates[l];
        GemmParams p = gemm_params_plain(lin, P + L.wih[l], gates, (int)rows2, 8 * H, in, in, in, 8 * H);
        p.bias = P + L.bsum[l];
        TRY(launch_gemm_f32(p, s, "train_lstm_inproj_gemm"));
        TRY(launch_lstm_rec(gates, P + L.whh[l], saved + S.hout[l], gates, saved + S.cst[l], B, Tp, H, s));
        lin = saved + S.hout[l];
        in = 2 * H;
        if (l < d.L - 1) {
            const float* mk = mask_lstm_host ? mask_lstm_host[l] : nullptr;
            if (mk) {
                ProfScope prof("train_elementwise", s, 0.0, (double)rows2 * 2 * H * 12);
                hipLaunchKernelGGL(mul_kernel, dim3(ew_blocks(rows2 * 2 * H / 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(lin),
                                   reinterpret_cast<const float4*>(mk), reinterpret_cast<float4*>(saved + S.hdrop[l]), rows2 * 2 * H / 4);
                RSAF_CHECK_HIP(hipGetLastError());
                lin = saved + S.hdrop[l];
            }
        }
    }
    // ---- attention pooling + dropout + classifier (:187-191) --------------------------------------------------------------
    {
        ProfScope prof("train_attnpool", s, 0.0, (double)rows2 * 2 * H * 8);
        if (H == 128)
            hipLaunchKernelGGL(attnpool_train_kernel<4>, dim3(B), dim3(ATT_WAVES * 64), 0, s, lin, P + L.watt, P + L.batt, saved + S.prob, saved + S.ctx, Tp);
        else
            hipLaunchKernelGGL(attnpool_train_kernel<2>, dim3(B), dim3(ATT_WAVES * 64), 0, s, lin, P + L.watt, P + L.batt, saved + S.prob, saved + S.ctx, Tp);
        hipLaunchKernelGGL(fc_fwd_kernel, dim3(B), dim3(256), 0, s, saved + S.ctx, mask_fc, P + L.wfc, P + L.bfc, logits, 2 * H, d.NC);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    if (bn_stats_out) RSAF_CHECK_HIP(hipMemcpyAsync(bn_stats_out, st, sizeof(float) * 5 * 3 * C, hipMemcpyDeviceToDevice, s));
    return RSAF_OK;
}

int rsaf_cnnlstm_train_backward(const float* x, int B, int T, int input_dim, int channels, int hidden, int num_classes,
                                int lstm_layers, int act, const float* params, const float* mask_block1,
                                const float* mask_block2, const float* const* mask_lstm_host, const float* mask_fc, float* saved,
                                int64_t saved_floats, float* scratch, int64_t scratch_floats, const float* dlogits, float* grads,
                                rsaf_stream_t stream) {
    Dims d{input_dim, channels, hidden, num_classes, lstm_layers, act};
    TRY(check_dims(d));
    RSAF_CHECK_ARG(B >= 1 && B <= 65535 && T >= 2, "bad batch / sequence length");
    RSAF_CHECK_ARG((int64_t)B * T <= 0x3fffffffLL, "B*T too large");
    RSAF_CHECK_ARG(x && params && saved && scratch && dlogits && grads, "NULL pointer");
    const SLayout S = make_slayout(d, B, T);
    const WLayout WL = make_wlayout(d, B, T);
    if (saved_floats < S.total || scratch_floats < WL.total) {
        set_error("rsaf_cnnlstm_train_backward: saved/scratch buffer too small");
        return RSAF_ERR_WORKSPACE;
    }
    const PLayout L = make_playout(d);
    hipStream_t s = (hipStream_t)stream;
    Ctx c{d, s, scratch, WL};
    const int D = d.D, C = d.C, H = d.H, Tp = T / 2, F = 2 * H;
    const int64_t rows = (int64_t)B * T, rows2 = (int64_t)B * Tp;
    const float* P = params;
    float* G = grads;
    float* st = saved + S.stat;
    auto stat = [&](int i) { return st + (int64_t)i * 3 * C; };
    float* bufA = scratch + WL.bufA;
    float* bufB = scratch + WL.bufB;
    float* bufC = scratch + WL.bufC;
    float* small = scratch + WL.small;
    float* dctx = small + 2 * 1024 + 64;               // after the BN `sums` area
    float* dwatt_part = dctx + (int64_t)B * F;
    float* dbatt_part = dwatt_part + (int64_t)B * F;
    float* dp_scr = dbatt_part + B + 4;

    // ---- classifier + attention pooling ----------------------------------------------------------------------------
    const float* seq_top = saved + S.hout[d.L - 1];
    float* dseq = bufA;                                 // [rows2][2H]
    {
        ProfScope prof("train_attnpool", s, 0.0, (double)rows2 * F * 12);
        hipLaunchKernelGGL(fc_bwd_kernel, dim3((F + 255) / 256), dim3(256), 0, s, dlogits, saved + S.ctx, mask_fc, P + L.wfc, G + L.wfc,
                           G + L.bfc, dctx, B, F, d.NC);
        if (H == 128)
            hipLaunchKernelGGL(attn_bwd_kernel<4>, dim3(B), dim3(ATT_WAVES * 64), 0, s, seq_top, saved + S.prob, dctx, P + L.watt, dp_scr, dseq,
                               dwatt_part, dbatt_part, Tp);
        else
            hipLaunchKernelGGL(attn_bwd_kernel<2>, dim3(B), dim3(ATT_WAVES * 64), 0, s, seq_top, saved + S.prob, dctx, P + L.watt, dp_scr, dseq,
                               dwatt_part, dbatt_part, Tp);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    TRY(colsum(c, dwatt_part, F, B, F, G + L.watt));
    TRY(colsum(c, dbatt_part, 1, B, 1, G + L.batt));

    // ---- LSTM layers, top down -----------------------------------------------------------------------------------------
    float* dcur = dseq;                                 // gradient w.r.t. the output of layer l  [rows2][2H]
    float* dnext = bufB;
    for (int l = d.L - 1; l >= 0; --l) {
        float* gates = saved + S.gates[l];
        const int in = l == 0 ? C : 2 * H;
        const float* lin = l == 0 ? saved + S.r2
                                  : ((mask_lstm_host && mask_lstm_host[l - 1]) ? saved + S.hdrop[l - 1] : saved + S.hout[l - 1]);
        {
            ProfScope prof("lstm_bwd_recurrent", s, 2.0 * B * Tp * 2.0 * 4 * H * H, 0.0);
            static const int small_max = [] { const char* e = getenv("RSAF_LSTM_SMALL_MAX"); return e ? atoi(e) : 1024; }();
            if (B <= small_max) {
                dim3 grid((B + 3) / 4, 2);
                if (H == 128) hipLaunchKernelGGL(lstm_bwd4_kernel<128>, grid, dim3(512), 0, s, gates, saved + S.cst[l], dcur, P + L.whh[l], B, Tp);
                else hipLaunchKernelGGL(lstm_bwd4_kernel<64>, grid, dim3(256), 0, s, gates, saved + S.cst[l], dcur, P + L.whh[l], B, Tp);
            } else {
                dim3 grid((B + 15) / 16, 2);
                const size_t lds = (size_t)2 * 16 * (4 * H + 4) * sizeof(float);
                if (H == 128)
                    RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_bwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                if (H == 128) hipLaunchKernelGGL(lstm_bwd_kernel<128>, grid, dim3(512), lds, s, gates, saved + S.cst[l], dcur, P + L.whh[l], B, Tp);
                else hipLaunchKernelGGL(lstm_bwd_kernel<64>, grid, dim3(256), lds, s, gates, saved + S.cst[l], dcur, P + L.whh[l], B, Tp);
            }
            RSAF_CHECK_HIP(hipGetLastError());
        }
        // gates now holds dgates (pre-activation gradients) [rows2][8H]
        TRY(colsum(c, gates, 8 * H, rows2, 8 * H, G + L.bsum[l]));
        TRY(wgrad(c, gates, 8 * H, 8 * H, lin, in, in, 1, 0, B, Tp, G + L.wih[l]));
        for (int dir = 0; dir < 2; ++dir)
            TRY(wgrad(c, gates + dir * 4 * H, 8 * H, 4 * H, saved + S.hout[l] + dir * H, 2 * H, H, 1, dir ? +1 : -1, B, Tp,
                      G + L.whh[l] + (int64_t)dir * 4 * H * H));
        // input gradient: dgates . W_ih  (B operand [K = 8H][N = in])
        {
            GemmParams p = gemm_params_plain(gates, P + L.wih[l], dnext, (int)rows2, in, 8 * H, 8 * H, in, in);
            p.b_kn = 1;
            TRY(launch_gemm_f32(p, s, "train_dgrad_gemm"));
        }
        if (l > 0 && mask_lstm_host && mask_lstm_host[l - 1]) {
            ProfScope prof("train_elementwise", s, 0.0, (double)rows2 * F * 12);
            hipLaunchKernelGGL(mul_kernel, dim3(ew_blocks(rows2 * F / 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(dnext),
                               reinterpret_cast<const float4*>(mask_lstm_host[l - 1]), reinterpret_cast<float4*>(dnext), rows2 * F / 4);
            RSAF_CHECK_HIP(hipGetLastError());
        }
        std::swap(dcur, dnext);
    }
    // dcur = dr2 [rows2][C]; the other of (bufA, bufB) is free
    float* dr2 = dcur;
    float* dz2 = dnext;
    // ---- res_block2 -------------------------------------------------------------------------------------------------------
    {
        ProfScope prof("train_elementwise", s, 0.0, (double)rows2 * C * 12);
        hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(rows2 * C)), dim3(256), 0, s, dr2, saved + S.z2, dz2, act, rows2 * C);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    float* dy = dr2;                                    // reuse
    TRY(bn_backward(c, dz2, nullptr, false, saved + S.y4, stat(4), P + L.c4.g, P + L.c4.be, rows2, G + L.c4.g, G + L.c4.be, dy));
    TRY(colsum(c, dy, C, rows2, C, G + L.c4.b));
    TRY(wgrad(c, dy, C, C, saved + S.a3d, C, C, 3, 0, B, Tp, G + L.c4.w));
    float* da = bufC;
    TRY(conv3_dgrad(c, dy, P + L.c4.w, da, B, Tp, C, C));
    TRY(bn_backward(c, da, mask_block2, true, saved + S.y3, stat(3), P + L.c3.g, P + L.c3.be, rows2, G + L.c3.g, G + L.c3.be, dy));
    TRY(colsum(c, dy, C, rows2, C, G + L.c3.b));
    TRY(wgrad(c, dy, C, C, saved + S.p, C, C, 3, 0, B, Tp, G + L.c3.w));
    TRY(conv3_dgrad(c, dy, P + L.c3.w, da, B, Tp, C, C));
    float* dpool = dy;                                  // dp = dgrad + dz2 (identity shortcut)
    {
        ProfScope prof("train_elementwise", s, 0.0, (double)rows2 * C * 12);
        hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(rows2 * C / 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(da),
                           reinterpret_cast<const float4*>(dz2), reinterpret_cast<float4*>(dpool), rows2 * C / 4);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    // ---- max_pool1d backward + activation backward of res_block1 ------------------------------------------------------------
    float* dz1 = bufC;                                  // [rows][C]  (da is dead)
    {
        ProfScope prof("train_elementwise", s, 0.0, (double)rows * C * 10);
        hipLaunchKernelGGL(pool_bwd_act_kernel, dim3(ew_blocks(rows * C)), dim3(256), 0, s, dpool, saved + S.z1, dz1, act, B, T, Tp, C);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    // ---- res_block1 -------------------------------------------------------------------------------------------------------
    float* dy1 = dz2 == bufA ? bufA : bufB;             // any buffer other than dz1 (bufC)
    float* dy2 = dy1 == bufA ? bufB : bufA;
    if (D != C) {
        TRY(bn_backward(c, dz1, nullptr, false, saved + S.ysc, stat(1), P + L.sc.g, P + L.sc.be, rows, G + L.sc.g, G + L.sc.be, dy1));
        TRY(colsum(c, dy1, C, rows, C, G + L.sc.b));
        TRY(wgrad(c, dy1, C, C, x, D, D, 1, 0, B, T, G + L.sc.w));
    }
    TRY(bn_backward(c, dz1, nullptr, false, saved + S.y2, stat(2), P + L.c2.g, P + L.c2.be, rows, G + L.c2.g, G + L.c2.be, dy2));
    TRY(colsum(c, dy2, C, rows, C, G + L.c2.b));
    TRY(wgrad(c, dy2, C, C, saved + S.a1d, C, C, 3, 0, B, T, G + L.c2.w));
    TRY(conv3_dgrad(c, dy2, P + L.c2.w, dy1, B, T, C, C));           // dy1 now holds d(a1d)
    TRY(bn_backward(c, dy1, mask_block1, true, saved + S.y1, stat(0), P + L.c1.g, P + L.c1.be, rows, G + L.c1.g, G + L.c1.be, dy2));
    TRY(colsum(c, dy2, C, rows, C, G + L.c1.b));
    TRY(wgrad(c, dy2, C, C, x, D, D, 3, 0, B, T, G + L.c1.w));
    return RSAF_OK;
}

}  // extern "C"
