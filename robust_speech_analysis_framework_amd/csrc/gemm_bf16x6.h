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
};

int launch_gemm_bf16x6(const Gemm6Params& p, hipStream_t stream, const char* tag);
// src[n] fp32 -> planes[0..2][n] (planes plane_stride elements apart)
int launch_split_bf16x3(const float* src, int64_t n, uint16_t* planes, int64_t plane_stride, hipStream_t stream);

}  // namespace rsaf
