// fp32-accurate GEMM on the fp16 matrix pipe of gfx950 from two-way operand splits (see gemm_f16x3.hip).
#pragma once
#include "gemm_f32.h"

namespace rsaf {

// C[z][m][n] = act( alpha * sum_k A[z][m][k] * B[n][k] + bias[n] + R[z][m][n] ).
// Every operand is two fp16 planes (hi, lo) of the fp32 value TIMES A POWER OF TWO chosen per row (or per batch), so
// that the row's largest magnitude sits below 2^15: x * s = hi + lo (+ at most 2^-22 |x s|).  The scales are exact, so
// they leave the product untouched: the epilogue multiplies by 1 / (s_a[m] s_b[n]).
struct GemmH3Params {
    const uint16_t* A;       // plane 0 (hi) of [M][lda] fp16 bit patterns; plane 1 (lo) a_plane elements later
    int64_t a_plane, lda, sA;   // sA: batch stride (elements)
    const float* a_scale;    // forward scale of A: a_scale[z1 * a_scale_zs + m * a_scale_ms]
    int a_scale_zs, a_scale_ms;
    const uint16_t* B;       // plane 0 of [N][ldb] (shared by every batch, or per group: sB2)
    int64_t b_plane, ldb;
    const float* b_scale;    // [N] forward scale of every weight row (+ z2 * sBias2 for groups)
    float* C;                // fp32 output [M][ldc] or nullptr
    int64_t ldc, sC;
    uint16_t* Cp;            // output as two fp16 planes (the next GEMM's A) or nullptr
    int64_t c_plane, ldcp, sCp;
    const float* c_scale;    // forward scale of the plane output: c_scale[z1 * c_scale_zs + m * c_scale_ms] (must bound the
    int c_scale_zs, c_scale_ms;   // output: |x| * scale < 65504; the callers derive it from a Cauchy-Schwarz bound)
    unsigned* amax_out;      // optional: atomicMax of the bit pattern of |x| (x = what is written) into
    int amax_zs;             //   amax_out[z1 * amax_zs + (amax_row_slot ? amax_row_slot[m] : 0)]
    const int* amax_row_slot;
    int amax_col_min;        // only columns >= amax_col_min take part (the V third of a fused q/k/v projection)
    const float* bias;       // [N] or nullptr
    const float* R;          // fp32 residual [M][ldr] or nullptr
    int64_t ldr, sR;
    int M, N, K, nz;
    // optional, ragged windows of one launch: ztab[2 z1] = rows of batch z1 (<= M; tiles past it are skipped),
    // ztab[2 z1 + 1] = element offset of the batch's fp32 output (packed windows), or -1 for z1 * sC
    const int64_t* ztab;
    int act;
    float alpha;
    int group_m;             // row-tiles per L2 group of the tile order (0 = default)
    // k16-panel layout of a plane of R rows: element (r, k) at (k / 16) * (R * 16) + r * 16 + k % 16, i.e. K / 16 panels of
    // [R][16]: the 32 rows x 32 bytes one DMA instruction moves are 1 KiB contiguous.  a_panel: A (R = M; lda, sA unused,
    // nz = 1); b_panel: B (R = N or b_panel_rows); cp_panel: the plane output laid out as the A of the next GEMM (R = M, its K = N).
    int a_panel, b_panel, cp_panel;
    // Two-level batches (grouped convolution: z = z1 * nz2 + z2, z2 = group): A += z1 sA + z2 sA2, B += z2 sB2,
    // C += z1 sC + z2 sC2, bias / b_scale += z2 sBias2.  nz2 <= 1: one level.
    int nz2;
    int64_t sA2, sB2, sC2, sBias2;
    int b_panel_rows;
    // A as a padded panel image shared by the taps of a convolution: a_panel with R = a_panel_rows rows per panel (batches
    // allowed: sA = rows between the batches' first rows x 16), and, with a_tap_panels = P > 0, K = taps x P panels where
    // k-tile kt reads panel kt % P one row further down per tap kt / P (a_tap_inv is filled in by the launcher)
    int a_panel_rows, a_tap_panels, a_tap_inv;
};

int launch_gemm_f16x3(const GemmH3Params& p, hipStream_t stream, const char* tag);

// Per-row statistics of a row-major fp32 matrix [rows][K] (ld elements between rows): scale[r] = the power of two that puts
// the row's largest magnitude into [2^14, 2^15) (1 for an all-zero row); optional: norm2[r] = the row's Euclidean norm,
// stat_max[0] = max over rows of the norm, stat_max[1] = max |element| (bit patterns, atomicMax: zero them first).
int launch_f16x2_row_scales(const float* src, int64_t rows, int K, int64_t ld, float* scale, float* norm2, unsigned* stat_max,
                            hipStream_t stream);
// fp32 [rows][K] (row-major, ld) times scale[r * scale_stride] -> two fp16 planes, row-major [rows][K] or k16 panels
int launch_split_f16x2(const float* src, int64_t rows, int K, int64_t ld, const float* scale, int scale_stride, uint16_t* planes,
                       int64_t plane_stride, int panels, hipStream_t stream);
// scale[i] = power of two with (bound_i) * scale in [2^14, 2^15), bound_i = amax_bits[i] (as float) * factor + add, where
// factor = factor_host * (factor_dev ? *factor_dev : 1) and add = add_dev ? *add_dev : 0
int launch_scale_from_bound(const unsigned* amax_bits, int64_t n, const float* factor_dev, float factor_host, const float* add_dev,
                            float* scale, hipStream_t stream);

// ---- device helpers shared with the producers that write plane operands (w2v2.hip) -------------------------------------
// the power of two s with bound * s in [2^14, 2^15) (bound > 0, finite); 1 for bound == 0
__host__ __device__ inline float f16x2_scale_for_bound(float bound) {
    union { float f; unsigned u; } v;
    v.f = bound;
    const int e = (int)((v.u >> 23) & 0xff);             // biased exponent: bound in [2^(e-127), 2^(e-126))
    if (e == 0) return 1.0f;                             // zero (or fp32 subnormal): nothing to scale
    int se = 127 + 14 - (e - 127);                       // biased exponent of the scale
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    v.u = (unsigned)se << 23;
    return v.f;
}
// 1 / s of a power of two, exactly
__host__ __device__ inline float pow2_inverse(float s) {
    union { float f; unsigned u; } v;
    v.f = s;
    v.u = 0x7F000000u - v.u;
    return v.f;
}

#if defined(__HIPCC__)
// GELU(x) = x / 2 (1 + erf(x / sqrt 2)) of TWO values at once on the packed-fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: one
// instruction, two elements).  erf: the two-interval form of the single-precision libm routines (|a| <= 0.9277: a + a P(a^2);
// beyond: 1 - exp2(Q(|a|)), log2(e) folded into Q's coefficients so that the exponential is the bare v_exp_f32), both intervals
// evaluated and selected per element: 18 packed + 10 scalar instructions per pair against ~40 per ELEMENT for the library
// erff with its divergent branches.  Error of GELU: 9.0e-8 of max(|x|, 1e-3) over [-8, 8] (the library form: 1.1e-7);
// replayed in emulated fp32 FMA arithmetic against float64 in tests/test_gelu_host.py.
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2 gelu_pair(gelu_f32x2 x) {
    const gelu_f32x2 a = x * 0.70710678118654752440f;
    gelu_f32x2 t;
    t.x = __builtin_fabsf(a.x); t.y = __builtin_fabsf(a.y);
    const gelu_f32x2 s = a * a;
    auto K = [](float v) { return gelu_f32x2{v, v}; };
    gelu_f32x2 r = __builtin_elementwise_fma(K(-2.4937484340625815e-05f), t, K(5.528365727514029e-04f));
    const gelu_f32x2 u = __builtin_elementwise_fma(K(-5.603376310318708e-03f), t, K(3.4992024302482605e-02f));
    r = __builtin_elementwise_fma(r, s, u);
    r = __builtin_elementwise_fma(r, t, K(-1.540479063987732e-01f));
    r = __builtin_elementwise_fma(r, t, K(-9.158901572227478e-01f));
    r = __builtin_elementwise_fma(r, t, K(-1.8570011854171753e-01f));
    r = __builtin_elementwise_fma(r, t, t * -1.4426950216293335f);
    gelu_f32x2 e;
    e.x = __builtin_amdgcn_exp2f(r.x); e.y = __builtin_amdgcn_exp2f(r.y);
    gelu_f32x2 far = K(1.0f) - e;
    far.x = __builtin_copysignf(far.x, a.x); far.y = __builtin_copysignf(far.y, a.y);
    gelu_f32x2 q = __builtin_elementwise_fma(K(-5.96761703e-4f), s, K(4.99119423e-3f));
    q = __builtin_elementwise_fma(q, s, K(-2.67681349e-2f));
    q = __builtin_elementwise_fma(q, s, K(1.12819925e-1f));
    q = __builtin_elementwise_fma(q, s, K(-3.76125336e-1f));
    q = __builtin_elementwise_fma(q, s, K(1.28379166e-1f));
    const gelu_f32x2 near = __builtin_elementwise_fma(q, a, a);
    gelu_f32x2 er;
    er.x = t.x > 0.927734375f ? far.x : near.x;
    er.y = t.y > 0.927734375f ? far.y : near.y;
    const gelu_f32x2 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, er, hx);
}
#endif

}  // namespace rsaf
