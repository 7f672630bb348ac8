// Exact-fp32 MFMA GEMM building block (gfx950, v_mfma_f32_32x32x2_f32).
//
//   C[z][m][n] = act( alpha * sum_k A[z][m][k] * B[z][.][.] + bias[n] + R[z][m][n] )
//
// A is row-major, K contiguous.  B is either [N,K] (K contiguous: PyTorch Linear/Conv weights,
// "NT") or [K,N] (N contiguous: attention P.V, "KN").  z = z1*nz2 + z2 selects a batch through two
// stride levels (chunk, head).  A 1-D convolution over a channels-last sequence is the same GEMM
// with lda = stride*Cin and K = taps*Cin; `a_pad_k` adds the zero padding of a k=3/pad=1 conv
// (first row masks k < a_pad_k, last row masks k >= K - a_pad_k) so the input is read in place.
#pragma once
#include "rsaf_common.h"

namespace rsaf {

enum Act : int { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2 };

struct GemmParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // [N] or nullptr
    const float* R;      // residual, same indexing as C (own ld/strides) or nullptr
    int M, N, K;
    int64_t lda, ldb, ldc, ldr;
    int nz, nz2;         // number of batches, inner batch count
    int64_t sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
    int64_t sBias2;      // bias offset per inner batch index z2 (grouped convolutions)
    int a_pad_k;
    int act;
    float alpha;
    int b_kn;            // 0: B[N][K]; 1: B[K][N]
};

// Enqueue the GEMM on `stream`.  `tag` names the kernel family for rsaf_prof_*.
int launch_gemm_f32(const GemmParams& p, hipStream_t stream, const char* tag);

inline GemmParams gemm_params_plain(const float* A, const float* B, float* C, int M, int N, int K,
                                    int64_t lda, int64_t ldb, int64_t ldc) {
    GemmParams p{};
    p.A = A; p.B = B; p.C = C; p.bias = nullptr; p.R = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = 0;
    p.nz = 1; p.nz2 = 1;
    p.sA1 = p.sA2 = p.sB1 = p.sB2 = p.sC1 = p.sC2 = p.sR1 = p.sR2 = 0;
    p.sBias2 = 0; p.a_pad_k = 0; p.act = ACT_NONE; p.alpha = 1.0f; p.b_kn = 0;
    return p;
}

}  // namespace rsaf
