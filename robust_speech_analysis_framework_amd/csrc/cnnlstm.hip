// CNN-LSTM-with-attention classifier forward for gfx950 (reference: CNNLSTM.forward,
// src/models.py:161-193, eval mode; harness behaviours of src/dl_cv_strategies.py:81-84 apply:
// zero-padded frames are NOT masked anywhere).
//
//   x[B,T,D] (read in its native layout: the permute of :172 costs nothing)
//   -> res_block1: act(conv3+BN) -> conv3+BN (+ conv1x1+BN shortcut) -> act     (fp32 MFMA GEMMs,
//      BN folded into the weights at load time, k=3 padding handled in the A-tile loader)
//   -> max_pool1d(2) -> res_block2 (identity shortcut)
//   -> 2-layer bidirectional LSTM: input projections for all time steps as one GEMM per layer
//      (N = 8H: both directions), then a persistent recurrent kernel
//   -> attention pooling (online softmax over time) + dropout(identity) + Linear -> logits[B,2]
//
// Recurrent kernel: one workgroup = 16 batch rows of one direction for all T' steps (rows are
// independent, so there is no inter-workgroup traffic).  H/16 waves; wave w owns hidden units
// 16w..16w+15 for all four gates, and keeps its slice of W_hh permanently in registers as
// v_mfma_f32_16x16x4_f32 B-fragments (H registers per lane).  Per step: the accumulators start from
// the prefetched input projection, h_{t-1} is read from LDS as the A operand, the gate nonlinearity
// and cell update are lane-local (all four gates of a (row, unit) pair land in the same lane and
// register index), h_t goes to LDS (double-buffered, one barrier per step) and to HBM.
#include <algorithm>
#include <cstdlib>

#include "cnnlstm_kernels.h"
#include "gemm_f32.h"
#include "gemm_f16x3.h"

namespace rsaf {
namespace cnnlstm {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct Dims {
    int D, C, H, NC, L, act;
};

static inline int64_t pad4(int64_t n) { return (n + 3) & ~int64_t(3); }

// ---- weight blob layout (floats); every segment starts 16-byte aligned -----------------------
struct Layout {
    int64_t w1, b1, wsc, bsc, w2, b2, w3, b3, w4, b4;
    int64_t wih[4], bih[4], whh[4];
    int64_t watt, batt, wfc, bfc, total;
};

static Layout make_layout(const Dims& d) {
    Layout L{};
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t s = o; o += pad4(n); return s; };
    L.w1 = take((int64_t)d.C * 3 * d.D);
    L.b1 = take(d.C);
    if (d.D != d.C) { L.wsc = take((int64_t)d.C * d.D); L.bsc = take(d.C); } else { L.wsc = L.bsc = -1; }
    L.w2 = take((int64_t)d.C * 3 * d.C); L.b2 = take(d.C);
    L.w3 = take((int64_t)d.C * 3 * d.C); L.b3 = take(d.C);
    L.w4 = take((int64_t)d.C * 3 * d.C); L.b4 = take(d.C);
    for (int l = 0; l < d.L; ++l) {
        const int in = l == 0 ? d.C : 2 * d.H;
        L.wih[l] = take((int64_t)8 * d.H * in);
        L.bih[l] = take(8 * d.H);
        L.whh[l] = take((int64_t)2 * 4 * d.H * d.H);
    }
    L.watt = take(2 * d.H); L.batt = take(1);
    L.wfc = take((int64_t)d.NC * 2 * d.H); L.bfc = take(d.NC);
    L.total = o;
    return L;
}

static int check_dims(const Dims& d) {
    RSAF_CHECK_ARG(d.D > 0 && d.D % 4 == 0, "input_dim must be a positive multiple of 4");
    RSAF_CHECK_ARG(d.C > 0 && d.C % 4 == 0, "cnn_out_channels must be a positive multiple of 4");
    RSAF_CHECK_ARG(d.H == 64 || d.H == 128, "lstm_hidden_dim must be 64 or 128 (reference search space)");
    RSAF_CHECK_ARG(d.NC >= 1 && d.NC <= 16, "num_classes must be in [1, 16]");
    RSAF_CHECK_ARG(d.L >= 1 && d.L <= 4, "lstm_layers must be in [1, 4]");
    RSAF_CHECK_ARG(d.act == ACT_GELU || d.act == ACT_SILU, "activation must be gelu (1) or silu (2)");
    return RSAF_OK;
}

// ---- max_pool1d(kernel_size=2) over time, channels-last ----------------------------------------
__global__ __launch_bounds__(256) void pool2_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                    int B, int T, int Tp, int C4) {
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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- persistent bidirectional LSTM recurrence ------------------------------------------------------
// SAVE (training): the post-activation gates overwrite the input projection in place (the lane that read an element
// is the lane that rewrites it, one step after the read of the next step was issued) and the cell state is kept.
template <int H, bool SAVE>
__global__ __launch_bounds__(H / 16 * 64) void lstm_rec_kernel(const float* xproj, const float* __restrict__ whh,
                                                               float* __restrict__ hout, float* gates_save,
                                                               float* __restrict__ c_save, int B, int T) {
    constexpr int LDH = H + 4;
    constexpr int KG = H / 16;                   // k-groups of 16 (4 MFMA k-steps each)
    __shared__ __attribute__((aligned(16))) float hbuf[2][16][LDH];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int col = lane & 15;
    const int q = lane >> 4;
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * 16;
    const int unit = 16 * w + col;

    // W_hh slice as B fragments: element (g, j) of gate gt holds W[gt*H + unit][16g + 4q + j]
    float breg[4][KG * 4];
    const float* wd = whh + (int64_t)dir * 4 * H * H;
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(wd + (int64_t)(gt * H + unit) * H + 16 * g + 4 * q);
            breg[gt][4 * g + 0] = v.x; breg[gt][4 * g + 1] = v.y;
            breg[gt][4 * g + 2] = v.z; breg[gt][4 * g + 3] = v.w;
        }
    for (int i = tid; i < 2 * 16 * LDH; i += H / 16 * 64) (&hbuf[0][0][0])[i] = 0.0f;

    // rows of this lane in the C/D map: q*4 + r, clamped to the last batch row (copies replicate it bit for bit)
    int64_t xoff[4], hoff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + q * 4 + r;
        const int bc = b < B ? b : B - 1;
        xoff[r] = (int64_t)bc * T * 8 * H + dir * 4 * H + unit;
        hoff[r] = (int64_t)bc * T * 2 * H + dir * H + unit;
    }
    float cst[4] = {0.f, 0.f, 0.f, 0.f};
    float xin[4][4];
    {
        const int t = dir ? T - 1 : 0;
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int r = 0; r < 4; ++r) xin[gt][r] = xproj[xoff[r] + (int64_t)t * 8 * H + gt * H];
    }
    vmem_drain();
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        f32x4 acc[4];
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[gt][r] = xin[gt][r];
        {                                          // prefetch the next step's input projection (the last step reloads
            const int tn = s + 1 < T ? (dir ? t - 1 : t + 1) : t;      // its own: the loop body stays branch-free)
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int r = 0; r < 4; ++r) xin[gt][r] = xproj[xoff[r] + (int64_t)tn * 8 * H + gt * H];
        }
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const float4 a4 = *reinterpret_cast<const float4*>(&hbuf[cur][col][16 * g + 4 * q]);
            const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt)
                    acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], breg[gt][4 * g + j], acc[gt], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = sigmoidf_(acc[0][r]);
            const float fg = sigmoidf_(acc[1][r]);
            const float gg = tanhf(acc[2][r]);
            const float og = sigmoidf_(acc[3][r]);
            cst[r] = fg * cst[r] + ig * gg;
            const float hv = og * tanhf(cst[r]);
            hbuf[cur ^ 1][q * 4 + r][unit] = hv;
            // rows past B are clamped to row B-1 and replicate it bit for bit: their stores rewrite the same values, and
            // leaving them unconditional keeps the loop branch-free (exact vmcnt bookkeeping, no store drain per step)
            hout[hoff[r] + (int64_t)t * 2 * H] = hv;
            if (SAVE) {
                float* gs = gates_save + xoff[r] + (int64_t)t * 8 * H;
                gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
                c_save[hoff[r] + (int64_t)t * 2 * H] = cst[r];
            }
        }
        lds_barrier();
        cur ^= 1;
    }
}

// ---- the same recurrence for small batches: 4 batch rows per workgroup -------------------------------------
// With a handful of sequences (the reference trains with batches of 4 and 8, src/dl_cv_strategies.py:234,265) the
// 16-row tile above wastes 3/4 of its MFMA work and leaves the step time at 16 rows' worth.  Here a workgroup owns 4
// rows of one direction and uses v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per instruction): row i
// of every block is batch row i, the 64 (block, j) columns of wave w are the four gates of hidden units 16w..16w+15
// (column c = 16*gate + u).  W_hh stays register-resident (H registers per lane: the lane's column over all k), h
// is broadcast from LDS.  The gate pre-activations of one (row, unit) land in four lanes (one per gate); they are
// exchanged through a wave-private LDS tile so that lane (q, u) updates cell (row q, unit u).
template <int H, bool SAVE>
__global__ __launch_bounds__(H / 16 * 64) void lstm_rec4_kernel(const float* xproj, const float* __restrict__ whh,
                                                                float* __restrict__ hout, float* gates_save,
                                                                float* __restrict__ c_save, int B, int T) {
    constexpr int LDH = H + 4;
    constexpr int NW = H / 16;
    __shared__ __attribute__((aligned(16))) float hbuf[2][4][LDH];
    __shared__ float gx[NW][4][80];                 // [wave][row][16*gate + u], row stride 80: conflict-free both ways
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int gq = lane >> 4, u = lane & 15;        // as a column: gate gq of unit u; as a cell owner: row gq of unit u
    const int dir = blockIdx.y, b0 = blockIdx.x * 4;
    const int unit = 16 * w + u;

    float breg[H];
    {
        const float* wr = whh + (int64_t)dir * 4 * H * H + (int64_t)(gq * H + unit) * H;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(wr + k);
            breg[k] = v.x; breg[k + 1] = v.y; breg[k + 2] = v.z; breg[k + 3] = v.w;
        }
    }
    for (int i = tid; i < 2 * 4 * LDH; i += NW * 64) (&hbuf[0][0][0])[i] = 0.0f;

    // as a column: input projections of rows 0..3; as a cell owner: outputs of row gq
    int64_t xoff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int bc = min(b0 + r, B - 1);
        xoff[r] = (int64_t)bc * T * 8 * H + dir * 4 * H + gq * H + unit;
    }
    const int bo = min(b0 + gq, B - 1);
    const int64_t hoff = (int64_t)bo * T * 2 * H + dir * H + unit;
    const int64_t goff = (int64_t)bo * T * 8 * H + dir * 4 * H + unit;
    const int arow = lane & 3;                      // A operand: lane 4*blk + i carries batch row i
    const float ya = gq == 2 ? 2.0f : 1.0f, yb = gq == 2 ? -2.0f : -1.0f, yc = gq == 2 ? -1.0f : 0.0f;
    float cst = 0.f;
    float xin[4];
    {
        const int t = dir ? T - 1 : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) xin[r] = xproj[xoff[r] + (int64_t)t * 8 * H];
    }
    vmem_drain();
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        f32x4 acc[4];
        acc[0] = f32x4{xin[0], xin[1], xin[2], xin[3]};
#pragma unroll
        for (int a = 1; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const int tn = s + 1 < T ? (dir ? t - 1 : t + 1) : t;      // branch-free: the last step reloads its own row
#pragma unroll
            for (int r = 0; r < 4; ++r) xin[r] = xproj[xoff[r] + (int64_t)tn * 8 * H];
        }
        // h in batches of 8 float4 (32 k), the next batch in flight while this one feeds the matrix pipe
        float4 ab[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ab[0][j] = *reinterpret_cast<const float4*>(&hbuf[cur][arow][4 * j]);
#pragma unroll
        for (int kb = 0; kb < H / 32; ++kb) {
            if (kb + 1 < H / 32) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    ab[(kb + 1) & 1][j] = *reinterpret_cast<const float4*>(&hbuf[cur][arow][32 * (kb + 1) + 4 * j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 a4 = ab[kb & 1][j];
                const int k = 32 * kb + 4 * j;
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.x, breg[k + 0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.y, breg[k + 1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.z, breg[k + 2], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.w, breg[k + 3], acc[3], 0, 0, 0);
            }
        }
        // gate nonlinearity in the column's lane (one gate type per lane: y = ya / (1 + exp(yb * x)) + yc is the
        // logistic function for i, f, o and tanh for g), then the 4x4 exchange through the wave's LDS tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pre = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
            gx[w][r][lane] = ya * __builtin_amdgcn_rcpf(1.0f + __expf(yb * pre)) + yc;
        }
        __builtin_amdgcn_wave_barrier();
        const float ig = gx[w][gq][u], fg = gx[w][gq][16 + u], gg = gx[w][gq][32 + u], og = gx[w][gq][48 + u];
        cst = fg * cst + ig * gg;
        const float hv = og * (2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * cst)) - 1.0f);
        hbuf[cur ^ 1][gq][unit] = hv;
        hout[hoff + (int64_t)t * 2 * H] = hv;          // rows past B replicate row B-1 (see lstm_rec_kernel)
        if (SAVE) {
            float* gs = gates_save + goff + (int64_t)t * 8 * H;
            gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
            c_save[hoff + (int64_t)t * 2 * H] = cst;
        }
        lds_barrier();
        cur ^= 1;
    }
}

// ---- attention pooling (softmax over time) + final Linear -----------------------------------------
template <int NF>   // features per lane: 2H = 64*NF
__global__ __launch_bounds__(256) void attnpool_fc_kernel(const float* __restrict__ seq, const float* __restrict__ watt,
                                                          const float* __restrict__ batt, const float* __restrict__ wfc,
                                                          const float* __restrict__ bfc, float* __restrict__ logits,
                                                          float* __restrict__ pooled_out, int T, int NC) {
    constexpr int F = 64 * NF;
    __shared__ float s_m[4], s_l[4], s_ctx[4][F], s_red[4][16];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const float* sb = seq + (int64_t)b * T * F;
    float wa[NF], ctx[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) { wa[i] = watt[lane + 64 * i]; ctx[i] = 0.f; }
    const float ba = batt[0];
    float m = -INFINITY, l = 0.f;
    for (int t = w; t < T; t += 4) {
        float v[NF];
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NF; ++i) { v[i] = sb[(int64_t)t * F + lane + 64 * i]; d += v[i] * wa[i]; }
        const float sc = wave_sum(d) + ba;
        const float mn = fmaxf(m, sc);
        const float a = expf(m - mn);
        const float p = expf(sc - mn);
        l = l * a + p;
#pragma unroll
        for (int i = 0; i < NF; ++i) ctx[i] = ctx[i] * a + p * v[i];
        m = mn;
    }
    if (lane == 0) { s_m[w] = m; s_l[w] = l; }
#pragma unroll
    for (int i = 0; i < NF; ++i) s_ctx[w][lane + 64 * i] = ctx[i];
    __syncthreads();
    const float M = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    float Lt = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) Lt += s_l[k] * expf(s_m[k] - M);
    // thread f (< F) owns pooled feature f
    float pooled = 0.f;
    const int f = threadIdx.x;
    if (f < F) {
#pragma unroll
        for (int k = 0; k < 4; ++k) pooled += s_ctx[k][f] * expf(s_m[k] - M);
        pooled /= Lt;
        if (pooled_out) pooled_out[(int64_t)b * F + f] = pooled;
    }
    if (!wfc) return;                                  // pooling only (AttentionPooling.forward, src/models.py:94-107)
    for (int c = 0; c < NC; ++c) {
        float part = (f < F) ? pooled * wfc[(int64_t)c * F + f] : 0.f;
        part = wave_sum(part);
        if (lane == 0) s_red[w][c] = part;
    }
    __syncthreads();
    if (threadIdx.x < NC)
        logits[(int64_t)b * NC + threadIdx.x] =
            s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x] +
            bfc[threadIdx.x];
}

}  // namespace cnnlstm

int launch_lstm_rec(const float* xproj, const float* whh, float* hout, float* gates_save, float* c_save, int B, int T,
                    int H, hipStream_t s) {
    using namespace cnnlstm;
    RSAF_CHECK_ARG(H == 64 || H == 128, "lstm_hidden_dim must be 64 or 128");
    RSAF_CHECK_ARG((gates_save == nullptr) == (c_save == nullptr), "gates_save and c_save go together");
    ProfScope prof("lstm_recurrent", s, 2.0 * B * T * 2.0 * 4 * H * H, 0.0);
    // 4-row workgroups while they still fit the chip about twice over; the 16-row tile beyond (same pipe time for 4x the rows)
    static const int small_max = [] { const char* e = getenv("RSAF_LSTM_SMALL_MAX"); return e ? atoi(e) : 1024; }();
    const bool small = B <= small_max;
    dim3 grid(small ? (B + 3) / 4 : (B + 15) / 16, 2);
#define RSAF_LSTM_LAUNCH(KERNEL, HH, SV) \
    hipLaunchKernelGGL((KERNEL<HH, SV>), grid, dim3(HH / 16 * 64), 0, s, xproj, whh, hout, gates_save, c_save, B, T)
    if (small) {
        if (gates_save) { if (H == 128) RSAF_LSTM_LAUNCH(lstm_rec4_kernel, 128, true); else RSAF_LSTM_LAUNCH(lstm_rec4_kernel, 64, true); }
        else { if (H == 128) RSAF_LSTM_LAUNCH(lstm_rec4_kernel, 128, false); else RSAF_LSTM_LAUNCH(lstm_rec4_kernel, 64, false); }
    } else {
        if (gates_save) { if (H == 128) RSAF_LSTM_LAUNCH(lstm_rec_kernel, 128, true); else RSAF_LSTM_LAUNCH(lstm_rec_kernel, 64, true); }
        else { if (H == 128) RSAF_LSTM_LAUNCH(lstm_rec_kernel, 128, false); else RSAF_LSTM_LAUNCH(lstm_rec_kernel, 64, false); }
    }
#undef RSAF_LSTM_LAUNCH
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

namespace cnnlstm {

static int conv3(const float* x, const float* wk, const float* bias, const float* R, int64_t ldr, int64_t sR,
                 float* y, int B, int T, int Cin, int Cout, int act, hipStream_t s) {
    GemmParams p = gemm_params_plain(x - Cin, wk, y, T, Cout, 3 * Cin, Cin, 3 * Cin, Cout);
    p.bias = bias; p.R = R; p.ldr = ldr; p.sR1 = sR;
    p.nz = B; p.nz2 = 1; p.sA1 = (int64_t)T * Cin; p.sC1 = (int64_t)T * Cout;
    p.a_pad_k = Cin; p.act = act;
    return launch_gemm_f32(p, s, "cnn_conv_gemm");
}


static int tap_copy(float* dst, const float* src, int64_t n, hipStream_t s) {
    if (dst) RSAF_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return RSAF_OK;
}

// ---- the same forward with the convolutions and the LSTM input projections on the fp16-split GEMM (gemm_f16x3.hip) -------
// Every GEMM operand is a pair of fp16 planes under a power-of-two scale.  A k = 3 / pad = 1 convolution reads three
// consecutive rows of a sequence, so its A operand is stored with one zero row in front of and behind every sequence
// ([B][T + 2][C]: the convolution is then a plain GEMM with lda = C, K = 3 C over the padded rows) and carries ONE scale per
// sequence: the input's exact maximum, or for a convolution's own output the bound sqrt(K) max|a| max_n |w_n|_2 + max|b|
// (+ max |residual|) with max|a| the largest |input| that the producing epilogue reported.
__device__ __forceinline__ void split2_c(float xs, unsigned short& h, unsigned short& l) {
    const _Float16 hh = (_Float16)xs;
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, (_Float16)(xs - (float)hh));
}

// largest |x| of every sequence (bit pattern of the float: non-negative floats order like unsigned integers)
__global__ __launch_bounds__(256) void seq_absmax_kernel(const float4* __restrict__ x, int64_t n4_per_seq, int chunks,
                                                         unsigned* __restrict__ amax) {
    const int z = blockIdx.y;
    const float4* p = x + (int64_t)z * n4_per_seq;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4_per_seq; i += (int64_t)chunks * 256) {
        const float4 v = p[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = wave_max_nonneg(m);
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(amax + z, __float_as_uint(m));
}

// scale[z] = the power of two for bound_z = amax_in[z] * factor_host * wstat[0] + bstat[1] (+ amax_res[z]); a NULL wstat = 1
__global__ __launch_bounds__(256) void cnn_scale_kernel(const unsigned* __restrict__ amax_in, const unsigned* __restrict__ amax_res,
                                                        const unsigned* __restrict__ wstat, const unsigned* __restrict__ bstat,
                                                        float factor_host, float* __restrict__ scale, int n) {
    const int z = blockIdx.x * 256 + threadIdx.x;
    if (z >= n) return;
    float b = __uint_as_float(amax_in[z]) * factor_host * (wstat ? __uint_as_float(wstat[0]) * 1.000001f : 1.0f);
    if (bstat) b += __uint_as_float(bstat[1]);
    if (amax_res) b += __uint_as_float(amax_res[z]);
    scale[z] = f16x2_scale_for_bound(b * 1.000001f);
}

// fp32 [B][T][C] -> plane pair [B][T + 2][C] (zero rows around every sequence), times scale[z]; POOL: the rows are first
// max-pooled in pairs (src is [B][2 T (+1)][C]) and the pooled fp32 rows are kept as well (the residual of res_block2)
template <bool POOL>
__global__ __launch_bounds__(256) void split_padded_kernel(const float4* __restrict__ src, int B, int T, int Tsrc, int C4,
                                                           const float* __restrict__ scale, unsigned short* __restrict__ planes,
                                                           int64_t plane, float4* __restrict__ pooled) {
    const int64_t n = (int64_t)B * (T + 2) * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const int64_t bt = i / C4;
        const int tp = (int)(bt % (T + 2));
        const int64_t b = bt / (T + 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tp >= 1 && tp <= T) {
            const int t = tp - 1;
            if (POOL) {
                const float4 a = src[(b * Tsrc + 2 * t) * C4 + c], d = src[(b * Tsrc + 2 * t + 1) * C4 + c];
                v = make_float4(fmaxf(a.x, d.x), fmaxf(a.y, d.y), fmaxf(a.z, d.z), fmaxf(a.w, d.w));
                pooled[(b * T + t) * C4 + c] = v;
            } else {
                v = src[(b * Tsrc + t) * C4 + c];
            }
        }
        const float sc = scale[b];
        unsigned short hh[4], ll[4];
        split2_c(v.x * sc, hh[0], ll[0]); split2_c(v.y * sc, hh[1], ll[1]);
        split2_c(v.z * sc, hh[2], ll[2]); split2_c(v.w * sc, hh[3], ll[3]);
        unsigned short* pp = planes + 4 * i;
        *reinterpret_cast<uint2*>(pp) = make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
        *reinterpret_cast<uint2*>(pp + plane) = make_uint2(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16));
    }
}

// fp32 [B][T][C] -> plane pair of the zero-padded sequences in k16 panels: [C / 16][B (T + 2)][16], times scale[z].  One
// wave moves 16 rows x 64 channels: a lane reads 16 channels of a row (64 bytes) and writes 32 bytes per plane; the 16 rows
// of a panel are 512 contiguous bytes.  The image is the A operand of BOTH the 3-tap convolution (a_tap_panels = C / 16:
// tap k of output row t is image row t + k) and the 1x1 shortcut (the centre rows).
__global__ __launch_bounds__(256) void split_padded_panels_kernel(const float4* __restrict__ src, int B, int T, int C,
                                                                  const float* __restrict__ scale, unsigned short* __restrict__ planes,
                                                                  int64_t plane) {
    const int64_t R = (int64_t)B * (T + 2);
    const int cgroups = C / 64;                                          // 64-channel groups per row (C % 64 == 0)
    const int64_t tiles = ((R + 15) / 16) * cgroups;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < tiles; tile += (int64_t)gridDim.x * 4) {
        const int cg = (int)(tile % cgroups);
        const int64_t r = (tile / cgroups) * 16 + (lane >> 2);
        if (r >= R) continue;
        const int c0 = cg * 64 + (lane & 3) * 16;
        const int64_t b = r / (T + 2);
        const int tp = (int)(r - b * (T + 2));
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tp >= 1 && tp <= T) {
            const float4* q = src + ((b * T + (tp - 1)) * C + c0) / 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = q[u];
        }
        const float sc = scale[b];
        unsigned hw[8], lw[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned short hh[4], ll[4];
            split2_c(v[u].x * sc, hh[0], ll[0]); split2_c(v[u].y * sc, hh[1], ll[1]);
            split2_c(v[u].z * sc, hh[2], ll[2]); split2_c(v[u].w * sc, hh[3], ll[3]);
            hw[2 * u] = hh[0] | ((unsigned)hh[1] << 16); hw[2 * u + 1] = hh[2] | ((unsigned)hh[3] << 16);
            lw[2 * u] = ll[0] | ((unsigned)ll[1] << 16); lw[2 * u + 1] = ll[2] | ((unsigned)ll[3] << 16);
        }
        unsigned short* pp = planes + ((int64_t)(c0 / 16) * R + r) * 16;
        *reinterpret_cast<uint4*>(pp) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *reinterpret_cast<uint4*>(pp + 8) = make_uint4(hw[4], hw[5], hw[6], hw[7]);
        *reinterpret_cast<uint4*>(pp + plane) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        *reinterpret_cast<uint4*>(pp + plane + 8) = make_uint4(lw[4], lw[5], lw[6], lw[7]);
    }
}

// fp32 [B][n4 float4] -> plane pair of the same shape, times scale[b]
__global__ __launch_bounds__(256) void split_rows_kernel(const float4* __restrict__ src, int B, int64_t n4_per_seq,
                                                         const float* __restrict__ scale, unsigned short* __restrict__ planes, int64_t plane) {
    const int64_t n = (int64_t)B * n4_per_seq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float4 v = src[i];
        const float sc = scale[i / n4_per_seq];
        unsigned short hh[4], ll[4];
        split2_c(v.x * sc, hh[0], ll[0]); split2_c(v.y * sc, hh[1], ll[1]);
        split2_c(v.z * sc, hh[2], ll[2]); split2_c(v.w * sc, hh[3], ll[3]);
        unsigned short* pp = planes + 4 * i;
        *reinterpret_cast<uint2*>(pp) = make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
        *reinterpret_cast<uint2*>(pp + plane) = make_uint2(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16));
    }
}

// zero the pad rows (first and last of every sequence) of a plane pair [B][T + 2][C] that a GEMM epilogue fills
__global__ __launch_bounds__(256) void zero_pad_rows_kernel(unsigned short* __restrict__ planes, int64_t plane, int B, int T, int C8) {
    const int64_t n = (int64_t)B * 2 * C8 * 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C8);
        int64_t r = i / C8;
        const int pl = (int)(r & 1); r >>= 1;
        const int which = (int)(r & 1);
        const int64_t b = r >> 1;
        const int64_t row = b * (T + 2) + (which ? T + 1 : 0);
        *reinterpret_cast<uint4*>(planes + pl * plane + (row * C8 + c) * 8) = make_uint4(0u, 0u, 0u, 0u);
    }
}

// widths the fp16-split GEMM takes (K and N multiples of 16); anything else (test geometries) stays on the exact-fp32 MFMA
static bool use_f16x3(int input_dim, int channels) {
    static const bool off = [] { const char* e = getenv("RSAF_CNN_F32"); return e && e[0] == '1'; }();
    return !off && input_dim % 16 == 0 && channels % 16 == 0;
}

struct F16Ws {          // float offsets into the workspace behind the fp32 buffers
    int64_t xp, c1p, poolp, c3p, c4p, hp, wp[5], ws[5], wih_p[4], wih_s[4], stat, amax, scale, one, total;
};
constexpr int CNN_NSTAT = 5 + 5 + 4;     // weight matrices w1 wsc w2 w3 w4, their biases, wih[l]
constexpr int CNN_NAMAX = 6, CNN_NSCALE = 6;

static F16Ws make_f16_ws(const Dims& d, int B, int T) {
    F16Ws w{};
    int64_t o = 0;
    auto take = [&](int64_t k) { int64_t s = o; o += pad4(k); return s; };
    const int Tp = T / 2;
    w.xp = take((int64_t)B * (T + 2) * d.D);
    w.c1p = take((int64_t)B * (T + 2) * d.C);
    w.poolp = take((int64_t)B * (Tp + 2) * d.C);
    w.c3p = take((int64_t)B * (Tp + 2) * d.C);
    w.c4p = take((int64_t)B * Tp * d.C);
    w.hp = take((int64_t)B * Tp * 2 * d.H);
    const int64_t wsz[5] = {(int64_t)d.C * 3 * d.D, (int64_t)d.C * d.D, (int64_t)d.C * 3 * d.C, (int64_t)d.C * 3 * d.C, (int64_t)d.C * 3 * d.C};
    for (int i = 0; i < 5; ++i) { w.wp[i] = take(wsz[i]); w.ws[i] = take(d.C); }
    for (int l = 0; l < d.L; ++l) { w.wih_p[l] = take((int64_t)8 * d.H * (l == 0 ? d.C : 2 * d.H)); w.wih_s[l] = take(8 * d.H); }
    w.stat = take(2 * CNN_NSTAT);
    w.amax = take((int64_t)CNN_NAMAX * B);
    w.scale = take((int64_t)CNN_NSCALE * B);
    w.one = take(4);
    w.total = o;
    return w;
}

static int forward_f16x3(const float* x, int B, int T, const Dims& d, const Layout& L, const float* W, float* ws, float* f16base,
                         float* logits, float* res1_out, float* res2_out, float* lstm_out, float* pooled_out, hipStream_t s) {
    int rc = RSAF_OK;
    const int D = d.D, C = d.C, H = d.H, Tp = T / 2;
    const int64_t conv = pad4((int64_t)B * T * C);
    float* bufA = ws;
    float* bufB = ws + conv;
    float* bufC = ws + 2 * conv;
    float* xproj = ws + 3 * conv;
    float* seq0 = xproj + pad4((int64_t)B * Tp * 8 * H);
    float* seq1 = seq0 + pad4((int64_t)B * Tp * 2 * H);
    const F16Ws F = make_f16_ws(d, B, T);
    auto planes = [&](int64_t off) { return reinterpret_cast<uint16_t*>(f16base + off); };
    unsigned* stat = reinterpret_cast<unsigned*>(f16base + F.stat);          // [CNN_NSTAT][2]: {max row norm, max |element|}
    unsigned* amax = reinterpret_cast<unsigned*>(f16base + F.amax);          // [6][B]: x, shortcut, conv1, conv2, conv3, (spare)
    float* scale = f16base + F.scale;                                       // [6][B]: x, conv1, pooled, conv3, conv4, (spare)
    float* one = f16base + F.one;
    enum { AX = 0, ASC = 1, AC1 = 2, AC2 = 3, AC3 = 4, AC4 = 5 };
    enum { SX = 0, SC1 = 1, SPOOL = 2, SC3 = 3, SC4 = 4 };
    RSAF_CHECK_HIP(hipMemsetAsync(stat, 0, sizeof(unsigned) * 2 * CNN_NSTAT, s));
    RSAF_CHECK_HIP(hipMemsetAsync(amax, 0, sizeof(unsigned) * CNN_NAMAX * B, s));
    // weights -> plane pairs in k16 panels with their row scales; statistics: matrix i at slot i, its bias at slot 5 + i
    const int64_t woff[5] = {L.w1, L.wsc, L.w2, L.w3, L.w4}, boff[5] = {L.b1, L.bsc, L.b2, L.b3, L.b4};
    const int wk[5] = {3 * D, D, 3 * C, 3 * C, 3 * C};
    for (int i = 0; i < 5; ++i) {
        if (woff[i] < 0) continue;                                          // no 1x1 shortcut when D == C
        if ((rc = launch_f16x2_row_scales(W + woff[i], C, wk[i], wk[i], f16base + F.ws[i], nullptr, stat + 2 * i, s))) return rc;
        if ((rc = launch_split_f16x2(W + woff[i], C, wk[i], wk[i], f16base + F.ws[i], 1, planes(F.wp[i]), (int64_t)C * wk[i], 1, s))) return rc;
        if ((rc = launch_f16x2_row_scales(W + boff[i], 1, C, C, one, nullptr, stat + 2 * (5 + i), s))) return rc;
    }
    for (int l = 0; l < d.L; ++l) {
        const int in = l == 0 ? C : 2 * H;
        if ((rc = launch_f16x2_row_scales(W + L.wih[l], 8 * H, in, in, f16base + F.wih_s[l], nullptr, stat + 2 * (10 + l), s))) return rc;
        if ((rc = launch_split_f16x2(W + L.wih[l], 8 * H, in, in, f16base + F.wih_s[l], 1, planes(F.wih_p[l]), (int64_t)8 * H * in, 1, s))) return rc;
    }
    RSAF_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(one), 0x46800000, 1, s));   // 16384.0f: |h| < 1 for every LSTM output
    auto cnn_scale = [&](int a_in, int a_res, int w_slot, int b_slot, float factor, int s_out) {
        hipLaunchKernelGGL(cnn_scale_kernel, dim3((B + 255) / 256), dim3(256), 0, s, amax + (int64_t)a_in * B,
                           a_res >= 0 ? amax + (int64_t)a_res * B : nullptr, w_slot >= 0 ? stat + 2 * w_slot : nullptr,
                           b_slot >= 0 ? stat + 2 * b_slot : nullptr, factor, scale + (int64_t)s_out * B, B);
    };
    // one GEMM of the block: A = padded plane pair (row stride lda, Tr rows per sequence, first tap at a_row0), per-sequence scale
    // (a_panels > 0: A is a panel image of a_panels channel panels over all B (T + 2) padded rows, see split_padded_panels_kernel)
    int a_panels = 0;
    auto gemm = [&](const uint16_t* A, int64_t a_plane, int64_t a_seq_rows, int a_row0, int lda, int s_in, int widx, int M, int N, int K,
                    float* Cf, uint16_t* Cp, int64_t c_plane, int64_t cp_seq_rows, int cp_row0, int s_out, const float* bias,
                    const float* R, int act, int amax_slot, const char* tag) {
        GemmH3Params p{};
        p.A = A + (int64_t)a_row0 * lda; p.a_plane = a_plane; p.lda = lda; p.sA = a_seq_rows * lda;
        if (a_panels > 0) {
            p.A = A + (int64_t)a_row0 * 16; p.lda = 16; p.sA = a_seq_rows * 16;
            p.a_panel = 1; p.a_panel_rows = (int)((int64_t)B * a_seq_rows);
            p.a_tap_panels = K > a_panels * 16 ? a_panels : 0;
        }
        p.a_scale = scale + (int64_t)s_in * B; p.a_scale_zs = 1; p.a_scale_ms = 0;
        p.B = planes(F.wp[widx]); p.b_plane = (int64_t)N * K; p.ldb = 16; p.b_panel = 1; p.b_scale = f16base + F.ws[widx];
        p.C = Cf; p.ldc = N; p.sC = (int64_t)M * N;
        p.Cp = Cp ? Cp + (int64_t)cp_row0 * N : nullptr; p.c_plane = c_plane; p.ldcp = N; p.sCp = cp_seq_rows * N;
        p.c_scale = s_out >= 0 ? scale + (int64_t)s_out * B : nullptr; p.c_scale_zs = 1; p.c_scale_ms = 0;
        p.amax_out = amax_slot >= 0 ? amax + (int64_t)amax_slot * B : nullptr; p.amax_zs = 1;
        p.bias = bias; p.R = R; p.ldr = N; p.sR = (int64_t)M * N;
        p.M = M; p.N = N; p.K = K; p.nz = B; p.act = act; p.alpha = 1.0f;
        return launch_gemm_f16x3(p, s, tag);
    };
    const int64_t pl_x = (int64_t)B * (T + 2) * D, pl_c = (int64_t)B * (T + 2) * C, pl_p = (int64_t)B * (Tp + 2) * C;
    // x -> padded planes under the exact maximum of every sequence; in k16 panels when the width allows (D % 64 == 0, image
    // rows below 2^27): row-major rows reach the DMA as 32-byte pieces of 512 different rows per k-tile, and with one column
    // tile (N = 128) nothing hides that (RSAF_CNN_ROWMAJOR=1: the row-major image, the A/B reference)
    const bool no_panels = [] { const char* e = getenv("RSAF_CNN_ROWMAJOR"); return e && e[0] == '1'; }();   // per call: the tests toggle it
    const bool x_panels = !no_panels && D % 64 == 0 && (int64_t)B * (T + 2) < ((int64_t)1 << 27);
    {
        ProfScope prof("cnn_split_input", s, 0.0, (double)B * T * D * 12.0);
        const int64_t n4 = (int64_t)T * D / 4;
        const int chunks = (int)std::min<int64_t>((n4 + 255) / 256, 64);
        hipLaunchKernelGGL(seq_absmax_kernel, dim3(chunks, B), dim3(256), 0, s, reinterpret_cast<const float4*>(x), n4, chunks, amax + AX * B);
        cnn_scale(AX, -1, -1, -1, 1.0f, SX);
        const int64_t tot = (int64_t)B * (T + 2) * (D / 4);
        if (x_panels) {
            const int64_t tiles = (((int64_t)B * (T + 2) + 15) / 16) * (D / 64);
            hipLaunchKernelGGL(split_padded_panels_kernel, dim3((unsigned)std::min<int64_t>((tiles + 3) / 4, 16384)), dim3(256), 0, s,
                               reinterpret_cast<const float4*>(x), B, T, D, scale + SX * B, planes(F.xp), pl_x);
        } else {
            hipLaunchKernelGGL(split_padded_kernel<false>, dim3((unsigned)std::min<int64_t>((tot + 255) / 256, 8192)), dim3(256), 0, s,
                               reinterpret_cast<const float4*>(x), B, T, T, D / 4, scale + SX * B, planes(F.xp), pl_x, nullptr);
        }
        RSAF_CHECK_HIP(hipGetLastError());
    }
    // res_block1 (src/models.py:64-76, :175): conv1 -> planes, shortcut -> fp32, conv2 (+ shortcut) -> fp32
    cnn_scale(AX, -1, 0, 5, sqrtf((float)(3 * D)) * 1.00001f, SC1);
    hipLaunchKernelGGL(zero_pad_rows_kernel, dim3(64), dim3(256), 0, s, planes(F.c1p), pl_c, B, T, C / 8);
    a_panels = x_panels ? D / 16 : 0;
    rc = gemm(planes(F.xp), pl_x, T + 2, 0, D, SX, 0, T, C, 3 * D, nullptr, planes(F.c1p), pl_c, T + 2, 1, SC1, W + L.b1, nullptr,
              d.act, AC1, "cnn_conv_gemm");
    if (rc) return rc;
    const float* sc = x;
    int sc_amax = AX;
    if (D != C) {
        rc = gemm(planes(F.xp), pl_x, T + 2, 1, D, SX, 1, T, C, D, bufB, nullptr, 0, 0, 0, -1, W + L.bsc, nullptr, ACT_NONE, ASC, "cnn_conv_gemm");
        if (rc) return rc;
        sc = bufB; sc_amax = ASC;
    }
    a_panels = 0;
    rc = gemm(planes(F.c1p), pl_c, T + 2, 0, C, SC1, 2, T, C, 3 * C, bufC, nullptr, 0, 0, 0, -1, W + L.b2, sc, d.act, AC2, "cnn_conv_gemm");
    if (rc) return rc;
    (void)sc_amax;
    if ((rc = tap_copy(res1_out, bufC, (int64_t)B * T * C, s))) return rc;
    // max_pool1d(2) (:177): pooled fp32 rows (the residual of res_block2) + padded planes; max |pooled| <= max |conv2 output|
    cnn_scale(AC2, -1, -1, -1, 1.0f, SPOOL);
    {
        ProfScope prof("cnn_pool2", s, 0.0, (double)B * T * C * 4 * 2.0);
        const int64_t tot = (int64_t)B * (Tp + 2) * (C / 4);
        hipLaunchKernelGGL(split_padded_kernel<true>, dim3((unsigned)std::min<int64_t>((tot + 255) / 256, 8192)), dim3(256), 0, s,
                           reinterpret_cast<const float4*>(bufC), B, Tp, T, C / 4, scale + SPOOL * B, planes(F.poolp), pl_p,
                           reinterpret_cast<float4*>(bufA));
        RSAF_CHECK_HIP(hipGetLastError());
    }
    // res_block2, identity shortcut (:178)
    cnn_scale(AC2, -1, 3, 8, sqrtf((float)(3 * C)) * 1.00001f, SC3);
    hipLaunchKernelGGL(zero_pad_rows_kernel, dim3(64), dim3(256), 0, s, planes(F.c3p), pl_p, B, Tp, C / 8);
    rc = gemm(planes(F.poolp), pl_p, Tp + 2, 0, C, SPOOL, 3, Tp, C, 3 * C, nullptr, planes(F.c3p), pl_p, Tp + 2, 1, SC3, W + L.b3, nullptr,
              d.act, AC3, "cnn_conv_gemm");
    if (rc) return rc;
    rc = gemm(planes(F.c3p), pl_p, Tp + 2, 0, C, SC3, 4, Tp, C, 3 * C, bufC, nullptr, 0, 0, 0, -1, W + L.b4, bufA, d.act, AC4, "cnn_conv_gemm");
    if (rc) return rc;
    if ((rc = tap_copy(res2_out, bufC, (int64_t)B * Tp * C, s))) return rc;
    // conv4's output as the plane pair of the first input projection, under the exact maximum of every sequence
    cnn_scale(AC4, -1, -1, -1, 1.0f, SC4);
    {
        const int64_t tot = (int64_t)B * Tp * (C / 4);
        hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)std::min<int64_t>((tot + 255) / 256, 8192)), dim3(256), 0, s,
                           reinterpret_cast<const float4*>(bufC), B, (int64_t)Tp * (C / 4), scale + SC4 * B, planes(F.c4p), (int64_t)B * Tp * C);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    // LSTM (:184): input projections on the same GEMM; layer 0 reads conv4's planes (one scale per sequence), the layers
    // behind it the plane pair of the previous layer's output under the fixed scale 2^14
    const float* lin = bufC;
    float* lout = seq0;
    for (int l = 0; l < d.L; ++l) {
        const int in = l == 0 ? C : 2 * H;
        GemmH3Params p{};
        p.B = planes(F.wih_p[l]); p.b_plane = (int64_t)8 * H * in; p.ldb = 16; p.b_panel = 1; p.b_scale = f16base + F.wih_s[l];
        p.C = xproj; p.ldc = 8 * H; p.bias = W + L.bih[l]; p.N = 8 * H; p.K = in; p.act = ACT_NONE; p.alpha = 1.0f;
        if (l == 0) {
            p.A = planes(F.c4p); p.a_plane = (int64_t)B * Tp * C; p.lda = C; p.sA = (int64_t)Tp * C;
            p.a_scale = scale + SC4 * B; p.a_scale_zs = 1; p.a_scale_ms = 0;
            p.M = Tp; p.nz = B; p.sC = (int64_t)Tp * 8 * H;
        } else {
            const int64_t rows = (int64_t)B * Tp;
            RSAF_CHECK_ARG(rows <= 0x7fffffffLL, "B*T too large");
            if ((rc = launch_split_f16x2(lin, rows, in, in, one, 0, planes(F.hp), rows * in, 1, s))) return rc;
            p.A = planes(F.hp); p.a_plane = rows * in; p.lda = 16; p.a_panel = 1;
            p.a_scale = one; p.a_scale_zs = 0; p.a_scale_ms = 0;
            p.M = (int)rows; p.nz = 1;
        }
        rc = launch_gemm_f16x3(p, s, "lstm_inproj_gemm");
        if (rc) return rc;
        rc = launch_lstm_rec(xproj, W + L.whh[l], lout, nullptr, nullptr, B, Tp, H, s);
        if (rc) return rc;
        lin = lout;
        lout = (lout == seq0) ? seq1 : seq0;
    }
    if ((rc = tap_copy(lstm_out, lin, (int64_t)B * Tp * 2 * H, s))) return rc;
    {
        ProfScope prof("attnpool_fc", s, 0.0, (double)B * Tp * 2 * H * 4);
        if (H == 128)
            hipLaunchKernelGGL(attnpool_fc_kernel<4>, dim3(B), dim3(256), 0, s, lin, W + L.watt, W + L.batt,
                               W + L.wfc, W + L.bfc, logits, pooled_out, Tp, d.NC);
        else
            hipLaunchKernelGGL(attnpool_fc_kernel<2>, dim3(B), dim3(256), 0, s, lin, W + L.watt, W + L.batt,
                               W + L.wfc, W + L.bfc, logits, pooled_out, Tp, d.NC);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    return RSAF_OK;
}

}  // namespace cnnlstm
}  // namespace rsaf

using namespace rsaf;
using namespace rsaf::cnnlstm;

extern "C" {

int64_t rsaf_cnnlstm_weight_floats(int input_dim, int channels, int hidden, int num_classes, int lstm_layers) {
    Dims d{input_dim, channels, hidden, num_classes, lstm_layers, ACT_SILU};
    if (check_dims(d) != RSAF_OK) return -1;
    return make_layout(d).total;
}

int rsaf_cnnlstm_weight_offsets(int input_dim, int channels, int hidden, int num_classes, int lstm_layers,
                                int64_t* offsets_host, int cap, int* n_host) {
    Dims d{input_dim, channels, hidden, num_classes, lstm_layers, ACT_SILU};
    int rc = check_dims(d);
    if (rc != RSAF_OK) return rc;
    RSAF_CHECK_ARG(offsets_host && n_host, "NULL output");
    const Layout L = make_layout(d);
    int64_t v[10 + 12 + 4];
    int n = 0;
    v[n++] = L.w1; v[n++] = L.b1; v[n++] = L.wsc; v[n++] = L.bsc; v[n++] = L.w2; v[n++] = L.b2;
    v[n++] = L.w3; v[n++] = L.b3; v[n++] = L.w4; v[n++] = L.b4;
    for (int l = 0; l < d.L; ++l) { v[n++] = L.wih[l]; v[n++] = L.bih[l]; v[n++] = L.whh[l]; }
    v[n++] = L.watt; v[n++] = L.batt; v[n++] = L.wfc; v[n++] = L.bfc;
    RSAF_CHECK_ARG(cap >= n, "offsets_host too small");
    for (int i = 0; i < n; ++i) offsets_host[i] = v[i];
    *n_host = n;
    return RSAF_OK;
}

int64_t rsaf_cnnlstm_workspace_bytes(int B, int T, int input_dim, int channels, int hidden, int lstm_layers) {
    if (B <= 0 || T < 2) return -1;
    const int64_t Tp = T / 2;
    const int64_t conv = pad4((int64_t)B * T * channels);
    const int64_t xp = pad4((int64_t)B * Tp * 8 * hidden);
    const int64_t sq = pad4((int64_t)B * Tp * 2 * hidden);
    int64_t f16 = 0;                                     // plane pairs, scales and statistics of the fp16-split GEMM path
    if (use_f16x3(input_dim, channels)) {
        Dims d{input_dim, channels, hidden, 2, lstm_layers, ACT_SILU};
        f16 = make_f16_ws(d, B, T).total;
    }
    return (3 * conv + xp + 2 * sq + f16) * (int64_t)sizeof(float);
}

static int forward_impl(const float* x, int B, int T, int input_dim, int channels, int hidden, int num_classes,
                        int lstm_layers, int act, const float* weights, void* workspace,
                        int64_t workspace_bytes, float* logits, float* res1_out, float* res2_out, float* lstm_out,
                        float* pooled_out, rsaf_stream_t stream) {
    Dims d{input_dim, channels, hidden, num_classes, lstm_layers, act};
    int rc = check_dims(d);
    if (rc != RSAF_OK) return rc;
    RSAF_CHECK_ARG(B >= 0 && B <= 65535, "batch must be in [0, 65535]");
    if (B == 0) return RSAF_OK;
    RSAF_CHECK_ARG(T >= 2, "sequence length must be >= 2 (max_pool1d(2) of the reference needs it)");
    RSAF_CHECK_ARG(x && weights && workspace && logits, "NULL pointer");
    const int64_t need = rsaf_cnnlstm_workspace_bytes(B, T, input_dim, channels, hidden, lstm_layers);
    if (workspace_bytes < need) {
        set_error("rsaf_cnnlstm_forward: workspace too small");
        return RSAF_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const Layout L = make_layout(d);
    const int D = d.D, C = d.C, H = d.H, Tp = T / 2;
    float* ws = static_cast<float*>(workspace);
    const int64_t conv = pad4((int64_t)B * T * C);
    float* bufA = ws;
    float* bufB = ws + conv;
    float* bufC = ws + 2 * conv;
    float* xproj = ws + 3 * conv;
    float* seq0 = xproj + pad4((int64_t)B * Tp * 8 * H);
    float* seq1 = seq0 + pad4((int64_t)B * Tp * 2 * H);
    const float* W = weights;
    if (use_f16x3(D, C))
        return forward_f16x3(x, B, T, d, L, W, ws, seq1 + pad4((int64_t)B * Tp * 2 * H), logits, res1_out, res2_out, lstm_out,
                             pooled_out, s);

    // res_block1 (src/models.py:64-76, :175)
    rc = conv3(x, W + L.w1, W + L.b1, nullptr, 0, 0, bufA, B, T, D, C, d.act, s);
    if (rc) return rc;
    const float* sc = x;
    int64_t ldsc = D;
    if (D != C) {
        GemmParams p = gemm_params_plain(x, W + L.wsc, bufB, T, C, D, D, D, C);
        p.bias = W + L.bsc; p.nz = B; p.sA1 = (int64_t)T * D; p.sC1 = (int64_t)T * C;
        rc = launch_gemm_f32(p, s, "cnn_conv_gemm");
        if (rc) return rc;
        sc = bufB; ldsc = C;
    }
    rc = conv3(bufA, W + L.w2, W + L.b2, sc, ldsc, (int64_t)T * ldsc, bufC, B, T, C, C, d.act, s);
    if (rc) return rc;
    if ((rc = tap_copy(res1_out, bufC, (int64_t)B * T * C, s))) return rc;
    // max_pool1d(2) (:177)
    {
        const int64_t n4 = (int64_t)B * Tp * (C / 4);
        const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 256 * 16);
        ProfScope prof("cnn_pool2", s, 0.0, (double)B * T * C * 4 * 1.5);
        hipLaunchKernelGGL(pool2_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(bufC),
                           reinterpret_cast<float4*>(bufA), B, T, Tp, C / 4);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    // res_block2, identity shortcut (:178)
    rc = conv3(bufA, W + L.w3, W + L.b3, nullptr, 0, 0, bufB, B, Tp, C, C, d.act, s);
    if (rc) return rc;
    rc = conv3(bufB, W + L.w4, W + L.b4, bufA, C, (int64_t)Tp * C, bufC, B, Tp, C, C, d.act, s);
    if (rc) return rc;
    if ((rc = tap_copy(res2_out, bufC, (int64_t)B * Tp * C, s))) return rc;
    // LSTM (:184)
    const float* lin = bufC;
    int in = C;
    float* lout = seq0;
    for (int l = 0; l < d.L; ++l) {
        const int64_t rows = (int64_t)B * Tp;
        RSAF_CHECK_ARG(rows <= 0x7fffffffLL, "B*T too large");
        GemmParams p = gemm_params_plain(lin, W + L.wih[l], xproj, (int)rows, 8 * H, in, in, in, 8 * H);
        p.bias = W + L.bih[l];
        rc = launch_gemm_f32(p, s, "lstm_inproj_gemm");
        if (rc) return rc;
        rc = launch_lstm_rec(xproj, W + L.whh[l], lout, nullptr, nullptr, B, Tp, H, s);
        if (rc) return rc;
        lin = lout; in = 2 * H;
        lout = (lout == seq0) ? seq1 : seq0;
    }
    if ((rc = tap_copy(lstm_out, lin, (int64_t)B * Tp * 2 * H, s))) return rc;
    // attention pooling + fc (:187-191)
    {
        ProfScope prof("attnpool_fc", s, 0.0, (double)B * Tp * 2 * H * 4);
        if (H == 128)
            hipLaunchKernelGGL(attnpool_fc_kernel<4>, dim3(B), dim3(256), 0, s, lin, W + L.watt, W + L.batt,
                               W + L.wfc, W + L.bfc, logits, pooled_out, Tp, d.NC);
        else
            hipLaunchKernelGGL(attnpool_fc_kernel<2>, dim3(B), dim3(256), 0, s, lin, W + L.watt, W + L.batt,
                               W + L.wfc, W + L.bfc, logits, pooled_out, Tp, d.NC);
        RSAF_CHECK_HIP(hipGetLastError());
    }
    return RSAF_OK;
}

int rsaf_cnnlstm_forward(const float* x, int B, int T, int input_dim, int channels, int hidden, int num_classes,
                         int lstm_layers, int act, const float* weights, void* workspace,
                         int64_t workspace_bytes, float* logits, rsaf_stream_t stream) {
    return forward_impl(x, B, T, input_dim, channels, hidden, num_classes, lstm_layers, act, weights, workspace,
                        workspace_bytes, logits, nullptr, nullptr, nullptr, nullptr, stream);
}

int rsaf_cnnlstm_forward_stages(const float* x, int B, int T, int input_dim, int channels, int hidden,
                                int num_classes, int lstm_layers, int act, const float* weights, void* workspace,
                                int64_t workspace_bytes, float* logits, float* res1_out, float* res2_out,
                                float* lstm_out, float* pooled_out, rsaf_stream_t stream) {
    return forward_impl(x, B, T, input_dim, channels, hidden, num_classes, lstm_layers, act, weights, workspace,
                        workspace_bytes, logits, res1_out, res2_out, lstm_out, pooled_out, stream);
}

int64_t rsaf_cnn_resblock_workspace_bytes(int B, int T, int out_channels) {
    if (B <= 0 || T < 1) return -1;
    return 2 * pad4((int64_t)B * T * out_channels) * (int64_t)sizeof(float);
}

int rsaf_cnn_resblock_forward(const float* x, int B, int T, int in_channels, int out_channels, int act,
                              const float* w1, const float* b1, const float* wsc, const float* bsc,
                              const float* w2, const float* b2, void* workspace, int64_t workspace_bytes,
                              float* y, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(in_channels > 0 && in_channels % 4 == 0 && out_channels > 0 && out_channels % 4 == 0,
                   "channel counts must be positive multiples of 4");
    RSAF_CHECK_ARG(act == ACT_GELU || act == ACT_SILU, "activation must be gelu (1) or silu (2)");
    RSAF_CHECK_ARG(B >= 0 && B <= 65535, "batch must be in [0, 65535]");
    if (B == 0) return RSAF_OK;
    RSAF_CHECK_ARG(T >= 1, "sequence length must be >= 1");
    RSAF_CHECK_ARG(x && w1 && b1 && w2 && b2 && workspace && y, "NULL pointer");
    RSAF_CHECK_ARG((wsc != nullptr) == (bsc != nullptr), "shortcut weight and bias come together");
    RSAF_CHECK_ARG(wsc || in_channels == out_channels, "identity shortcut needs in_channels == out_channels");
    if (workspace_bytes < rsaf_cnn_resblock_workspace_bytes(B, T, out_channels)) {
        set_error("rsaf_cnn_resblock_forward: workspace too small");
        return RSAF_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int D = in_channels, C = out_channels;
    float* bufA = static_cast<float*>(workspace);
    float* bufB = bufA + pad4((int64_t)B * T * C);
    int rc = conv3(x, w1, b1, nullptr, 0, 0, bufA, B, T, D, C, act, s);
    if (rc) return rc;
    const float* sc = x;
    int64_t ldsc = D;
    if (wsc) {
        GemmParams p = gemm_params_plain(x, wsc, bufB, T, C, D, D, D, C);
        p.bias = bsc; p.nz = B; p.sA1 = (int64_t)T * D; p.sC1 = (int64_t)T * C;
        rc = launch_gemm_f32(p, s, "cnn_conv_gemm");
        if (rc) return rc;
        sc = bufB; ldsc = C;
    }
    return conv3(bufA, w2, b2, sc, ldsc, (int64_t)T * ldsc, y, B, T, C, C, act, s);
}

int rsaf_attnpool_forward(const float* seq, int B, int T, int features, const float* watt, const float* batt,
                          float* pooled, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(features == 128 || features == 256, "features (2 * lstm_hidden_dim) must be 128 or 256");
    RSAF_CHECK_ARG(B >= 0 && T >= 1, "bad shape");
    if (B == 0) return RSAF_OK;
    RSAF_CHECK_ARG(seq && watt && batt && pooled, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("attnpool_fc", s, 0.0, (double)B * T * features * 4);
    if (features == 256)
        hipLaunchKernelGGL(attnpool_fc_kernel<4>, dim3(B), dim3(256), 0, s, seq, watt, batt,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr, pooled, T, 0);
    else
        hipLaunchKernelGGL(attnpool_fc_kernel<2>, dim3(B), dim3(256), 0, s, seq, watt, batt,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr, pooled, T, 0);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

}  // extern "C"
