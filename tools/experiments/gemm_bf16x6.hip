// EXPERIMENT (opt-in, not on the default path): fp32-accurate GEMM on the bf16 matrix pipe of gfx950.
//
// Every fp32 operand is split into three bf16 terms, a = a1 + a2 + a3 (each the bf16 rounding of what the previous
// terms left over: 3 x 8 significand bits = the 24 bits of a float), and the product is rebuilt from the six partial
// products whose weight is at least 2^-16 of the leading one,
//     a*b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are below 2^-24 |a b| each, i.e. the result
// differs from an fp32 FMA chain by a few units of fp32 rounding (measured against float64 in tools/gemm_bf16x6_bench.py)
// while the matrix pipe runs 16x faster per MAC: 6 instructions per 16 k against 8 instructions of 64 cycles.
// B (the weights) is split once by rsaf_split_bf16x3; A (the activations) is split on the fly while its k-tile moves
// from registers to LDS.
//
//   C[m][n] = act( alpha * sum_k A[m][k] * B[n][k] + bias[n] + R[m][n] ),   K % 32 == 0, plain row-major NT.
//
// 128 x 128 x 32 block tile, 4 waves of 64 x 64 (2 x 2 MFMA tiles); LDS image per operand: 3 planes of [128][32] bf16,
// 16-byte chunks XOR-swizzled with (row >> 1) & 3 so that the ds_read_b128 fragment reads are conflict-free.
#include <algorithm>
#include <cstdlib>

#include "gemm_f32.h"
#include "rsaf_exp.h"

namespace rsaf {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ float act_apply6(float v, int act) {
    if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    return v;
}

__device__ __forceinline__ unsigned short bf16_bits(float x) {
    const __bf16 h = (__bf16)x;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// a = h + m + l (+ at most 2^-24 |a|)
__device__ __forceinline__ void split3(float a, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = bf16_bits(a);
    const float r1 = a - bf16_to_f32(h);
    m = bf16_bits(r1);
    const float r2 = r1 - bf16_to_f32(m);
    l = bf16_bits(r2);
}

__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ src, int64_t n,
                                                           unsigned short* __restrict__ planes) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        unsigned short h, m, l;
        split3(src[i], h, m, l);
        planes[i] = h; planes[n + i] = m; planes[2 * n + i] = l;
    }
}

struct Gemm6Params {
    const float* A;
    const unsigned short* B;     // plane 0; planes are `plane_stride` elements apart, each [N][ldb]
    int64_t plane_stride;
    float* C;
    const float* bias;
    const float* R;
    int M, N, K;
    int64_t lda, ldb, ldc, ldr;
    int act;
    float alpha;
};

__global__ __launch_bounds__(256, 2) void gemm_bf16x6_kernel(const Gemm6Params p) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int PLANE = BM * BK;                       // bf16 elements per plane of one operand
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * PLANE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_n * tiles_m;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int grp = wg / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_grp = wg - grp * per_group;
    const int m0 = (first_m + in_grp % gsz) * BM;
    const int n0 = (in_grp / gsz) * BN;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // A: thread -> rows (tid >> 3) + 32 i, float4 column c4 = tid & 7 (k = 4 c4 .. 4 c4 + 3)
    const int c4 = tid & 7, ar0 = tid >> 3;
    const float* aptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + ar0 + 32 * i;
        aptr[i] = p.A + (int64_t)(gm < p.M ? gm : p.M - 1) * p.lda + 4 * c4;
    }
    // B: thread -> row tid >> 1, 16-byte chunks 2 (tid & 1) and 2 (tid & 1) + 1 of every plane
    const int brow = tid >> 1, bch = 2 * (tid & 1);
    const unsigned short* bptr;
    {
        const int gn = n0 + brow;
        bptr = p.B + (int64_t)(gn < p.N ? gn : p.N - 1) * p.ldb + 8 * bch;
    }
    float4 ra[4];
    uint4 rb[3][2];
    const int nk = p.K / BK;

#define G6_GLOAD(KT)                                                                                      \
    do {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4*>(aptr[i] + (KT) * BK); \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                \
            const uint4* s_ = reinterpret_cast<const uint4*>(bptr + pl * p.plane_stride + (KT) * BK);     \
            rb[pl][0] = s_[0]; rb[pl][1] = s_[1];                                                         \
        }                                                                                                 \
    } while (0)

#define G6_LSTORE()                                                                                       \
    do {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
            const int row = ar0 + 32 * i;                                                                 \
            unsigned short hh[4], mm[4], ll[4];                                                           \
            split3(ra[i].x, hh[0], mm[0], ll[0]); split3(ra[i].y, hh[1], mm[1], ll[1]);                   \
            split3(ra[i].z, hh[2], mm[2], ll[2]); split3(ra[i].w, hh[3], mm[3], ll[3]);                   \
            const int off = row * BK + (((c4 >> 1) ^ ((row >> 1) & 3)) << 3) + ((c4 & 1) << 2);           \
            *reinterpret_cast<uint2*>(&As[off]) = make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16)); \
            *reinterpret_cast<uint2*>(&As[PLANE + off]) = make_uint2(mm[0] | ((unsigned)mm[1] << 16), mm[2] | ((unsigned)mm[3] << 16)); \
            *reinterpret_cast<uint2*>(&As[2 * PLANE + off]) = make_uint2(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16)); \
        }                                                                                                 \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                  \
            _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                 \
                *reinterpret_cast<uint4*>(&Bs[pl * PLANE + brow * BK + (((bch + c) ^ ((brow >> 1) & 3)) << 3)]) = rb[pl][c]; \
    } while (0)

    G6_GLOAD(0);
    G6_LSTORE();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) G6_GLOAD(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2][3], bf[2][3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = wm0 + mt * 32 + l31;
                const int off = row * BK + (((2 * ks + h) ^ ((row >> 1) & 3)) << 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[mt][pl] = *reinterpret_cast<const bf16x8*>(&As[pl * PLANE + off]);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int row = wn0 + nt * 32 + l31;
                const int off = row * BK + (((2 * ks + h) ^ ((row >> 1) & 3)) << 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[nt][pl] = *reinterpret_cast<const bf16x8*>(&Bs[pl * PLANE + off]);
            }
            // smallest terms first
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x16 c = acc[mt][nt];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][2], bf[nt][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][1], bf[nt][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][0], bf[nt][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][1], bf[nt][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][0], bf[nt][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][0], bf[nt][0], c, 0, 0, 0);
                    acc[mt][nt] = c;
                }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            G6_LSTORE();
            __syncthreads();
        }
    }
#undef G6_GLOAD
#undef G6_LSTORE

#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int gn = n0 + wn0 + nt * 32 + l31;
        const int n_ok = gn < p.N;
        const int gnc = n_ok ? gn : 0;
        const float bv = p.bias ? p.bias[gnc] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int gm_base = m0 + wm0 + mt * 32 + 4 * h;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int gm = gm_base + (e & 3) + 8 * (e >> 2);
                const float rv = p.R ? p.R[(int64_t)(gm < p.M ? gm : 0) * p.ldr + gnc] : 0.0f;
                const float v = act_apply6(p.alpha * acc[mt][nt][e] + bv + rv, p.act);
                if (n_ok & (gm < p.M)) p.C[(int64_t)gm * p.ldc + gn] = v;
            }
        }
    }
}

// ---- both operands pre-split: k-tiles of 16 by LDS-DMA, double-buffered -------------------------------------------
// LDS image per stage: A planes then B planes, each [128 rows][16 bf16] = 32-byte rows, written linearly by
// global_load_lds (one wave-instruction = 1 KiB = 32 rows); the two 16-byte chunks of a row are swapped on the source
// side for rows with (row >> 2) & 1 so that the fragment reads (lane -> row l31, chunk lane >> 5) are conflict-free.
typedef __attribute__((address_space(3))) void* lds_ptr6;
typedef const __attribute__((address_space(1))) void* glb_ptr6;

struct Gemm6DmaParams {
    const unsigned short* A;     // plane 0 of [M][lda] bf16; planes a_plane elements apart
    const unsigned short* B;     // plane 0 of [N][ldb]
    int64_t a_plane, b_plane;
    float* C;
    const float* bias;
    const float* R;
    int M, N, K;
    int64_t lda, ldb, ldc, ldr;
    int act;
    float alpha;
};

__global__ __launch_bounds__(256, 2) void gemm_bf16x6_dma_kernel(const Gemm6DmaParams p) {
    constexpr int BM = 128, BN = 128, BK = 16;
    constexpr int PLANE = BM * BK;                       // 2048 bf16 = 4 KiB
    constexpr int STAGE = 6 * PLANE;                     // 24 KiB
    constexpr int NST = 3;                               // stages: DMA runs two k-tiles ahead of the multiply
    __shared__ __attribute__((aligned(1024))) unsigned short smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_n * tiles_m;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int grp = wg / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_grp = wg - grp * per_group;
    const int m0 = (first_m + in_grp % gsz) * BM;
    const int n0 = (in_grp / gsz) * BN;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // DMA: a plane of one operand is 4 wave-instructions of 32 rows; wave w issues instruction w of every plane.
    // lane -> row 32 w + (lane >> 1), LDS chunk lane & 1 <- global chunk (lane & 1) ^ ((row >> 2) & 1)
    const int drow = 32 * wave + (lane >> 1);
    const int dch = (lane & 1) ^ ((drow >> 2) & 1);
    const unsigned short* asrc;
    const unsigned short* bsrc;
    {
        const int gm = m0 + drow, gn = n0 + drow;
        asrc = p.A + (int64_t)(gm < p.M ? gm : p.M - 1) * p.lda + 8 * dch;
        bsrc = p.B + (int64_t)(gn < p.N ? gn : p.N - 1) * p.ldb + 8 * dch;
    }
    const int nk = p.K / BK;

#define G6_DMA(KT, ST)                                                                                          \
    do {                                                                                                        \
        unsigned short* st_ = smem + (ST) * STAGE + wave * 512;                                                 \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                      \
            __builtin_amdgcn_global_load_lds((glb_ptr6)(asrc + pl * p.a_plane + (KT) * BK), (lds_ptr6)(st_ + pl * PLANE), 16, 0, 0);       \
            __builtin_amdgcn_global_load_lds((glb_ptr6)(bsrc + pl * p.b_plane + (KT) * BK), (lds_ptr6)(st_ + (3 + pl) * PLANE), 16, 0, 0); \
        }                                                                                                       \
    } while (0)

    G6_DMA(0, 0);
    if (nk > 1) G6_DMA(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // stage (st + 2) % 3 was read in iteration kt - 1: every wave is past that iteration's barrier
        if (kt + 2 < nk) G6_DMA(kt + 2, st >= 1 ? st - 1 : 2);
        const unsigned short* img = smem + st * STAGE;
        bf16x8 af[2][3], bf[2][3];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = wm0 + mt * 32 + l31;
            const int off = row * BK + ((h ^ ((row >> 2) & 1)) << 3);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[mt][pl] = *reinterpret_cast<const bf16x8*>(&img[pl * PLANE + off]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int row = wn0 + nt * 32 + l31;
            const int off = row * BK + ((h ^ ((row >> 2) & 1)) << 3);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[nt][pl] = *reinterpret_cast<const bf16x8*>(&img[(3 + pl) * PLANE + off]);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x16 c = acc[mt][nt];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][2], bf[nt][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][1], bf[nt][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][0], bf[nt][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][1], bf[nt][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][0], bf[nt][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][0], bf[nt][0], c, 0, 0, 0);
                acc[mt][nt] = c;
            }
        // the next k-tile (issued one iteration ago: 6 DMA instructions per wave and k-tile) must have landed
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        st = st == NST - 1 ? 0 : st + 1;
    }
#undef G6_DMA

#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int gn = n0 + wn0 + nt * 32 + l31;
        const int n_ok = gn < p.N;
        const int gnc = n_ok ? gn : 0;
        const float bv = p.bias ? p.bias[gnc] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int gm_base = m0 + wm0 + mt * 32 + 4 * h;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int gm = gm_base + (e & 3) + 8 * (e >> 2);
                const float rv = p.R ? p.R[(int64_t)(gm < p.M ? gm : 0) * p.ldr + gnc] : 0.0f;
                const float v = act_apply6(p.alpha * acc[mt][nt][e] + bv + rv, p.act);
                if (n_ok & (gm < p.M)) p.C[(int64_t)gm * p.ldc + gn] = v;
            }
        }
    }
}

}  // namespace rsaf

using namespace rsaf;

extern "C" int rsaf_split_bf16x3(const float* src, int64_t n, uint16_t* planes, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(n >= 0, "negative length");
    if (n == 0) return RSAF_OK;
    RSAF_CHECK_ARG(src && planes, "NULL pointer");
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, n,
                       reinterpret_cast<unsigned short*>(planes));
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

extern "C" int rsaf_gemm_f32_bf16x6(const float* A, const uint16_t* B_planes, int64_t plane_stride, float* C,
                                    const float* bias, const float* R, int M, int N, int K, int64_t lda, int64_t ldb,
                                    int64_t ldc, int64_t ldr, int act, float alpha, rsaf_stream_t stream) {
    RSAF_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (M == 0 || N == 0) return RSAF_OK;
    RSAF_CHECK_ARG(A && B_planes && C, "NULL operand");
    RSAF_CHECK_ARG(K % 32 == 0 && K >= 32, "K must be a positive multiple of 32");
    RSAF_CHECK_ARG(lda % 4 == 0 && ldb % 8 == 0 && plane_stride % 8 == 0, "lda % 4, ldb % 8, plane_stride % 8 required");
    RSAF_CHECK_ARG((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B_planes) & 15) == 0,
                   "A and B must be 16-byte aligned");
    RSAF_CHECK_ARG(!R || ldr > 0, "residual needs ldr");
    RSAF_CHECK_ARG(act >= 0 && act <= 2, "act must be 0 (none), 1 (gelu) or 2 (silu)");
    Gemm6Params p{A, reinterpret_cast<const unsigned short*>(B_planes), plane_stride, C, bias, R, M, N, K, lda, ldb, ldc, ldr, act, alpha};
    const int tiles = ((N + 127) / 128) * ((M + 127) / 128);
    ProfScope prof("gemm_bf16x6", (hipStream_t)stream, 2.0 * M * (double)N * K, 0.0);
    hipLaunchKernelGGL(gemm_bf16x6_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

extern "C" int rsaf_gemm_bf16x6_presplit(const uint16_t* A_planes, int64_t a_plane_stride, const uint16_t* B_planes,
                                         int64_t b_plane_stride, float* C, const float* bias, const float* R, int M, int N,
                                         int K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int act, float alpha,
                                         rsaf_stream_t stream) {
    RSAF_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (M == 0 || N == 0) return RSAF_OK;
    RSAF_CHECK_ARG(A_planes && B_planes && C, "NULL operand");
    RSAF_CHECK_ARG(K % 16 == 0 && K >= 16, "K must be a positive multiple of 16");
    RSAF_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && a_plane_stride % 8 == 0 && b_plane_stride % 8 == 0,
                   "lda, ldb and the plane strides must be multiples of 8 elements");
    RSAF_CHECK_ARG((reinterpret_cast<uintptr_t>(A_planes) & 15) == 0 && (reinterpret_cast<uintptr_t>(B_planes) & 15) == 0,
                   "A and B must be 16-byte aligned");
    RSAF_CHECK_ARG(!R || ldr > 0, "residual needs ldr");
    RSAF_CHECK_ARG(act >= 0 && act <= 2, "act must be 0 (none), 1 (gelu) or 2 (silu)");
    Gemm6DmaParams p{reinterpret_cast<const unsigned short*>(A_planes), reinterpret_cast<const unsigned short*>(B_planes),
                     a_plane_stride, b_plane_stride, C, bias, R, M, N, K, lda, ldb, ldc, ldr, act, alpha};
    const int tiles = ((N + 127) / 128) * ((M + 127) / 128);
    ProfScope prof("gemm_bf16x6", (hipStream_t)stream, 2.0 * M * (double)N * K, 0.0);
    hipLaunchKernelGGL(gemm_bf16x6_dma_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}
