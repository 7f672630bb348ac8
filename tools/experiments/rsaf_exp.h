/* rsaf_exp.h - C ABI of the EXPERIMENTAL library tools/experiments/librsaf_exp.so (not part of librsaf.so, nothing in
 * the product path loads it).  See DESIGN.md "Experiment: fp32-accurate GEMM on the bf16 matrix pipe". */
#ifndef RSAF_EXP_H
#define RSAF_EXP_H
#include "rsaf.h"
#ifdef __cplusplus
extern "C" {
#endif
/* fp32-accurate GEMM on the bf16 matrix pipe
 * Same contraction as rsaf_gemm_f32 for the plain case (A [M,K] float32 row-major, B [N,K], K % 32 == 0, no batching,
 * no convolution padding), computed as six bf16 MFMA partial products of three-way bf16 splits of both operands with
 * fp32 accumulation (gemm_bf16x6.hip): error a few fp32 roundings, matrix pipe 16x faster per MAC.
 * rsaf_split_bf16x3 writes the three planes of a weight matrix ([3][n] bf16 bit patterns) once; B_planes points at
 * plane 0 of a [N][ldb] matrix, planes `plane_stride` elements apart.                                              */
int rsaf_split_bf16x3(const float* src, int64_t n, uint16_t* planes, rsaf_stream_t stream);
int rsaf_gemm_f32_bf16x6(const float* A, const uint16_t* B_planes, int64_t plane_stride, float* C,
                         const float* bias, const float* R, int M, int N, int K, int64_t lda, int64_t ldb,
                         int64_t ldc, int64_t ldr, int act, float alpha, rsaf_stream_t stream);
/* both operands pre-split (A_planes: [3][M][lda] bf16, K % 16 == 0): k-tiles by LDS-DMA, no conversion in the kernel */
int rsaf_gemm_bf16x6_presplit(const uint16_t* A_planes, int64_t a_plane_stride, const uint16_t* B_planes,
                              int64_t b_plane_stride, float* C, const float* bias, const float* R, int M, int N,
                              int K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int act, float alpha,
                              rsaf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
