// fp32-accurate GEMM on the bf16 matrix pipe of gfx950 (see gemm_bf16x6.hip).
#pragma once
#include "gemm_f32.h"

namespace rsaf {

// C[z][m][n] = act( alpha * sum_k A[z][m][k] * B[n][k] + bias[n] + R[z][m][n] ), operands as three bf16 planes each.
struct Gemm6Params {
    const uint16_t* A;       // plane 0 of [M][lda] bf16 bit patterns; planes a_plane elements apart
    int64_t a_plane, lda, sA;   // sA: batch stride (elements)
    const uint16_t* B;       // plane 0 of [N][ldb] (shared by every batch)
    int64_t b_plane, ldb;
    float* C;                // fp32 output [M][ldc] or nullptr
    int64_t ldc, sC;
    uint16_t* Cp;            // output as three bf16 planes (the next GEMM's A) or nullptr
    int64_t c_plane, ldcp, sCp;
    const float* bias;       // [N] or nullptr
    const float* R;          // fp32 residual [M][ldr] or nullptr
    int64_t ldr, sR;
    int M, N, K, nz;
    int act;
    float alpha;
    int group_m;             // row-tiles per L2 group of the tile order (0 = default)
    // k16-panel layout of a plane of R rows: element (r, k) at (k / 16) * (R * 16) + r * 16 + k % 16, i.e. K / 16 panels of
    // [R][16]: the 32 rows x 32 bytes one DMA instruction moves are 1 KiB contiguous (8 full 128-byte lines instead of
    // a quarter of 32 lines).  a_panel: A (R = M; lda, sA unused, nz = 1); b_panel: B (R = N; ldb unused);
    // cp_panel: the plane output, laid out as the A operand of the next GEMM (R = M, its K = N; ldcp unused).
    int a_panel, b_panel, cp_panel;
    // Two-level batches (grouped convolution: batch z = z1 * nz2 + z2, z2 = group): A += z1 sA + z2 sA2, B += z2 sB2,
    // C += z1 sC + z2 sC2, bias += z2 sBias2.  nz2 <= 1: one level (B and bias shared by every batch).  Only with the
    // fp32 output, without residual, A row-major.
    int nz2;
    int64_t sA2, sB2, sC2, sBias2;
    int b_panel_rows;        // b_panel: rows of the panels B lives in (0 = N); a group's rows are a slice of them
};

int launch_gemm_bf16x6(const Gemm6Params& p, hipStream_t stream, const char* tag);
// src[n] fp32 -> planes[0..2][n] (planes plane_stride elements apart)
int launch_split_bf16x3(const float* src, int64_t n, uint16_t* planes, int64_t plane_stride, hipStream_t stream);
// src[rows][K] fp32 (row-major) -> three planes in the k16-panel layout of `rows` rows (K % 16 == 0)
int launch_split_bf16x3_panels(const float* src, int64_t rows, int K, uint16_t* planes, int64_t plane_stride, hipStream_t stream);

}  // namespace rsaf
