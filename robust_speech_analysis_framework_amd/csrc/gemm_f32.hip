// Exact-fp32 MFMA GEMM for gfx950: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bitwise a
// k-ordered fmaf chain, 157.3 TFLOP/s dense peak).  bf16/fp8 MFMA would break the 1e-4 parity
// bar of the Wav2Vec2 / CNN-LSTM paths, so every dense contraction of the hot path runs here.
//
// Block tile BM x BN x 32, 256 threads = 4 waves, each wave a WM x WN tile of 32x32 MFMA tiles.
// A/B k-tiles go HBM -> registers (float4, 128-byte rows per 8 lanes) -> LDS with a row stride of
// 36 floats, which makes the ds_read_b128 fragment reads conflict-free (16 rows x 4 banks cover
// the 64 banks).  Each float4 fragment feeds four consecutive MFMA k-steps: inside an 8-wide
// k-group lanes 0-31 supply k = 0..3 and lanes 32-63 k = 4..7, for A and B alike, so every product
// still pairs equal k.  The next k-tile is prefetched into registers while the current one is
// multiplied.  Workgroup ids are remapped so that the 8 XCDs each own a contiguous range of tiles
// (tiles sharing an A row-panel hit the same L2).  The plain NT shapes with K % 32 == 0 (all Wav2Vec2 layers)
// take the LDS-DMA variant further down (global_load_lds, swizzled unpadded image): +2.5-6 % on those shapes.
#include <cstdlib>

#include "gemm_f32.h"

namespace rsaf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    return v;
}

template <int BM, int BN, int WM, int WN, bool BKN, bool DB>
__global__ __launch_bounds__(256, DB ? 2 : 3) void gemm_f32_kernel(const GemmParams p) {
    constexpr int BK = 32;
    constexpr int LDS_K = BK + 4;                 // 36-float rows (NT images)
    constexpr int LDB_N = BN + 4;                 // KN image row
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int A_IT = BM / 32;                 // float4 per thread per k-tile
    constexpr int B_IT = BN / 32;
    constexpr int A_FLOATS = BM * LDS_K;
    constexpr int B_FLOATS = BKN ? BK * LDB_N : BN * LDS_K;
    // DB: two LDS images, the next k-tile is written to the other image right after the multiply and
    // one barrier per k-tile remains (the image being overwritten was last read one barrier ago)
    __shared__ __attribute__((aligned(16))) float smem[(DB ? 2 : 1) * (A_FLOATS + B_FLOATS)];
    constexpr int STAGE = A_FLOATS + B_FLOATS;
    int wr_off = 0, rd_off = 0;                   // float offsets of the LDS image being written / read
#define As (smem + wr_off)
#define Bs (smem + wr_off + A_FLOATS)
#define AsR (smem + rd_off)
#define BsR (smem + rd_off + A_FLOATS)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // ---- tile coordinates (XCD-aware bijective remap of the 1-D grid) ----
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_n * tiles_m;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // grouped order: walk GROUP_M row-tiles x all column-tiles column by column, so the ~64 blocks an
    // XCD runs at once form an 8 x 8 patch that shares A row-panels AND B column-panels in its L2
    // (row-major order streamed every B panel from beyond L2: 58 % hit rate, 18x the algorithmic fetch)
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int grp = wg / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_grp = wg - grp * per_group;
    const int m0 = (first_m + in_grp % gsz) * BM;
    const int n0 = (in_grp / gsz) * BN;

    const int z = blockIdx.y;
    const int z1 = z / p.nz2, z2 = z - z1 * p.nz2;
    const float* __restrict__ A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const float* __restrict__ B = p.B + z1 * p.sB1 + z2 * p.sB2;
    float* __restrict__ C = p.C + z1 * p.sC1 + z2 * p.sC2;
    const float* __restrict__ R = p.R ? p.R + z1 * p.sR1 + z2 * p.sR2 : nullptr;
    const float* __restrict__ bias = p.bias ? p.bias + z2 * p.sBias2 : nullptr;

    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    float4 ra[A_IT], rb[B_IT];
    const int nk = (p.K + BK - 1) / BK;

    // ---- per-thread load coordinates, fixed over the K loop.  Everything below is branch-free:
    //      masked elements are loaded from an always-valid address and zeroed by a select (a branch
    //      around a load makes hipcc drain vmcnt(0) per element and serialises the prefetch).
    const int c4 = tid & 7;                       // float4 column inside the 32-wide k-tile
    const int r0 = tid >> 3;                      // rows r0 + 32*i
    const float* aptr[A_IT];
    int arow_ok[A_IT], alo[A_IT], ahi[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int gm = m0 + r0 + 32 * i;
        arow_ok[i] = gm < p.M;
        const int gmc = arow_ok[i] ? gm : p.M - 1;
        aptr[i] = A + (int64_t)gmc * p.lda;
        alo[i] = (p.a_pad_k > 0) & (gmc == 0);
        ahi[i] = (p.a_pad_k > 0) & (gmc == p.M - 1);
    }
    const float* bptr[B_IT];
    int b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        if constexpr (!BKN) {
            const int gn = n0 + r0 + 32 * i;
            b_ok[i] = gn < p.N;
            bptr[i] = B + (int64_t)(b_ok[i] ? gn : p.N - 1) * p.ldb;
        } else {
            const int idx = tid + 256 * i;
            const int gn = n0 + (idx % (BN / 4)) * 4;
            b_ok[i] = gn < p.N;
            bptr[i] = B + (b_ok[i] ? gn : 0);
        }
    }

// raw loads only: the zero-select is applied in RSAF_LSTORE (after the MFMAs of the current tile),
// so nothing consumes the prefetched registers before the multiply and the loads stay in flight
#define RSAF_GLOAD(KT)                                                                          \
    do {                                                                                        \
        const int gk = (KT) * BK + c4 * 4;                                                      \
        const int gkc = (gk < p.K) ? gk : 0;                                                    \
        /* k past the end (K % 32 != 0): any valid address; with a_pad_k the row base of the first row lies \
           a_pad_k floats in front of the buffer, so offset 0 would be out of bounds there */    \
        const int gka = (gk < p.K) ? gk : p.a_pad_k;                                            \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                      \
            const int hit = (alo[i] & (gk < p.a_pad_k)) | (ahi[i] & (gk >= p.K - p.a_pad_k));   \
            const int off = hit ? p.a_pad_k : gka;                                              \
            ra[i] = *reinterpret_cast<const float4*>(aptr[i] + off);                            \
        }                                                                                       \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                      \
            if constexpr (!BKN) {                                                               \
                rb[i] = *reinterpret_cast<const float4*>(bptr[i] + gkc);                        \
            } else {                                                                            \
                const int krow = (KT) * BK + (tid + 256 * i) / (BN / 4);                        \
                rb[i] = *reinterpret_cast<const float4*>(                                       \
                    bptr[i] + (int64_t)((krow < p.K) ? krow : 0) * p.ldb);                      \
            }                                                                                   \
        }                                                                                       \
    } while (0)

#define RSAF_LSTORE(KT)                                                                         \
    do {                                                                                        \
        const int gk = (KT) * BK + c4 * 4;                                                      \
        const int kok = gk < p.K;                                                               \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                      \
            const int hit = (alo[i] & (gk < p.a_pad_k)) | (ahi[i] & (gk >= p.K - p.a_pad_k));   \
            const int ok = kok & arow_ok[i] & (hit ^ 1);                                        \
            float4 v = ra[i];                                                                   \
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f; \
            *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * LDS_K + c4 * 4]) = v;                \
        }                                                                                       \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                      \
            float4 v = rb[i];                                                                   \
            if constexpr (!BKN) {                                                               \
                const int ok = kok & b_ok[i];                                                   \
                v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f; \
                *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * LDS_K + c4 * 4]) = v;            \
            } else {                                                                            \
                const int idx = tid + 256 * i;                                                  \
                const int ok = (((KT) * BK + idx / (BN / 4)) < p.K) & b_ok[i];                  \
                v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f; \
                *reinterpret_cast<float4*>(&Bs[(idx / (BN / 4)) * LDB_N + (idx % (BN / 4)) * 4]) = v; \
            }                                                                                   \
        }                                                                                       \
    } while (0)

    RSAF_GLOAD(0);
    RSAF_LSTORE(0);
    if constexpr (DB) {
        if (nk > 1) RSAF_GLOAD(1);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (DB) {
            // software pipeline, depth 2: registers hold tile kt+1 (loaded one iteration ago, so its
            // latency is long hidden) -> store it to the other LDS image, refill the registers with
            // tile kt+2, then multiply tile kt.  One barrier per k-tile; the image written here was
            // last read before the previous barrier.
            rd_off = (kt & 1) * STAGE;
            if (kt + 1 < nk) {
                wr_off = ((kt + 1) & 1) * STAGE;
                RSAF_LSTORE(kt + 1);
                if (kt + 2 < nk) RSAF_GLOAD(kt + 2);
            }
        } else {
            if (kt + 1 < nk) RSAF_GLOAD(kt + 1);
        }
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float af[TM][4], bf[TN][4];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const float4 v = *reinterpret_cast<const float4*>(
                    &AsR[(wm0 + mt * 32 + l31) * LDS_K + 8 * g + 4 * h]);
                af[mt][0] = v.x; af[mt][1] = v.y; af[mt][2] = v.z; af[mt][3] = v.w;
            }
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                if constexpr (!BKN) {
                    const float4 v = *reinterpret_cast<const float4*>(
                        &BsR[(wn0 + nt * 32 + l31) * LDS_K + 8 * g + 4 * h]);
                    bf[nt][0] = v.x; bf[nt][1] = v.y; bf[nt][2] = v.z; bf[nt][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bf[nt][j] = BsR[(8 * g + 4 * h + j) * LDB_N + wn0 + nt * 32 + l31];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bf[nt][j],
                                                                           acc[mt][nt], 0, 0, 0);
        }
        if constexpr (DB) {
            __syncthreads();
        } else {
            __syncthreads();
            if (kt + 1 < nk) {
                RSAF_LSTORE(kt + 1);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: alpha, bias, residual, activation; C/D map: col = lane&31,
    //      row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
        const int gn = n0 + wn0 + nt * 32 + l31;
        const int n_ok = gn < p.N;
        const int gnc = n_ok ? gn : 0;
        const float bv = bias ? bias[gnc] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int gm_base = m0 + wm0 + mt * 32 + 4 * h;
#pragma unroll
            for (int half = 0; half < 2; ++half) {        // 8 outputs at a time keeps the epilogue's
                float rv[8];                              // register footprint below the main loop's
                if (R) {   // uniform branch; the residual loads are issued together (clamped, unmasked)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = half * 8 + q;
                        const int gm = gm_base + (e & 3) + 8 * (e >> 2);
                        rv[q] = R[(int64_t)(gm < p.M ? gm : 0) * p.ldr + gnc];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) rv[q] = 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = half * 8 + q;
                    const int gm = gm_base + (e & 3) + 8 * (e >> 2);
                    const float v = act_apply(p.alpha * acc[mt][nt][e] + bv + rv[q], p.act);
                    if (n_ok & (gm < p.M)) C[(int64_t)gm * p.ldc + gn] = v;
                }
            }
        }
    }
}


// ---- LDS-DMA variant for the plain NT shapes (B[N,K], K % 32 == 0, no padded taps) --------------------------
// Same 128 x 128 x 32 tile and MFMA schedule, but the k-tiles go HBM -> LDS directly
// (global_load_lds_dwordx4: no staging registers, no ds_write).  One wave-instruction writes 1 KiB of LDS
// linearly (lane l -> base + 16 l), so the image is unpadded [row][8 chunks of 16 B]; the bank spread comes from
// an XOR swizzle chunk ^ ((row >> 1) & 7) applied to the per-lane SOURCE address and again to the fragment
// reads (16 consecutive rows of one chunk column then cover all 64 banks).  Tile kt+1 is issued before the
// multiply of tile kt into the other image; vmcnt(0) + barrier at the end of the k-step orders it for the reads.
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

template <int BM, int BN>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64, 512 / ((BM / 64) * (BN / 64) * 64)) void gemm_f32_glds_kernel(const GemmParams p) {
    constexpr int BK = 32, WM = 64, WN = 64, TM = 2, TN = 2, WAVES_N = BN / WN, NWAVES = (BM / WM) * WAVES_N;
    constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;
    constexpr int A_INS = BM / 8 / NWAVES, B_INS = BN / 8 / NWAVES;      // 1-KiB wave-instructions per wave per k-tile
    extern __shared__ __attribute__((aligned(1024))) float smem_dyn[];
    float* smem = smem_dyn;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
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

    const int z = blockIdx.y;
    const int z1 = z / p.nz2, z2 = z - z1 * p.nz2;
    const float* __restrict__ A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const float* __restrict__ B = p.B + z1 * p.sB1 + z2 * p.sB2;
    float* __restrict__ C = p.C + z1 * p.sC1 + z2 * p.sC2;
    const float* __restrict__ R = p.R ? p.R + z1 * p.sR1 + z2 * p.sR2 : nullptr;
    const float* __restrict__ bias = p.bias ? p.bias + z2 * p.sBias2 : nullptr;

    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // per-lane source pointers of this wave's DMA instructions (rows past the end re-read the last row: their
    // products land in accumulator rows / columns that are never stored)
    const float* asrc[A_INS];
    const float* bsrc[B_INS];
    const int prow = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int i = 0; i < A_INS; ++i) {
        const int row = 8 * (wave * A_INS + i) + prow;
        const int gm = m0 + row;
        asrc[i] = A + (int64_t)(gm < p.M ? gm : p.M - 1) * p.lda + 4 * (pc ^ ((row >> 1) & 7));
    }
#pragma unroll
    for (int i = 0; i < B_INS; ++i) {
        const int row = 8 * (wave * B_INS + i) + prow;
        const int gn = n0 + row;
        bsrc[i] = B + (int64_t)(gn < p.N ? gn : p.N - 1) * p.ldb + 4 * (pc ^ ((row >> 1) & 7));
    }
    const int nk = p.K / BK;

#define RSAF_DMA(KT, ST)                                                                                     \
    do {                                                                                                     \
        float* stage_ = smem + (ST) * STAGE;                                                                 \
        _Pragma("unroll") for (int i = 0; i < A_INS; ++i)                                                    \
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(asrc[i] + (KT) * BK),                            \
                                             (lds_void_ptr)(stage_ + (wave * A_INS + i) * 256), 16, 0, 0);   \
        _Pragma("unroll") for (int i = 0; i < B_INS; ++i)                                                    \
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(bsrc[i] + (KT) * BK),                            \
                                             (lds_void_ptr)(stage_ + A_FLOATS + (wave * B_INS + i) * 256), 16, 0, 0); \
    } while (0)

    RSAF_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fsw = (l31 >> 1) & 7;               // swizzle of this lane's fragment rows (tile bases are multiples of 32)
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) RSAF_DMA(kt + 1, (kt + 1) & 1);
        const float* a_img = smem + (kt & 1) * STAGE;
        const float* b_img = a_img + A_FLOATS;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float af[TM][4], bf[TN][4];
            const int ch = ((2 * g + h) ^ fsw) * 4;
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const float4 v = *reinterpret_cast<const float4*>(&a_img[(wm0 + mt * 32 + l31) * BK + ch]);
                af[mt][0] = v.x; af[mt][1] = v.y; af[mt][2] = v.z; af[mt][3] = v.w;
            }
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const float4 v = *reinterpret_cast<const float4*>(&b_img[(wn0 + nt * 32 + l31) * BK + ch]);
                bf[nt][0] = v.x; bf[nt][1] = v.y; bf[nt][2] = v.z; bf[nt][3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bf[nt][j],
                                                                           acc[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef RSAF_DMA

#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
        const int gn = n0 + wn0 + nt * 32 + l31;
        const int n_ok = gn < p.N;
        const int gnc = n_ok ? gn : 0;
        const float bv = bias ? bias[gnc] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int gm_base = m0 + wm0 + mt * 32 + 4 * h;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float rv[8];
                if (R) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = half * 8 + q;
                        const int gm = gm_base + (e & 3) + 8 * (e >> 2);
                        rv[q] = R[(int64_t)(gm < p.M ? gm : 0) * p.ldr + gnc];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) rv[q] = 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = half * 8 + q;
                    const int gm = gm_base + (e & 3) + 8 * (e >> 2);
                    const float v = act_apply(p.alpha * acc[mt][nt][e] + bv + rv[q], p.act);
                    if (n_ok & (gm < p.M)) C[(int64_t)gm * p.ldc + gn] = v;
                }
            }
        }
    }
}

template <int BM, int BN>
static int launch_glds_cfg(const GemmParams& p, hipStream_t s) {
    constexpr int THREADS = (BM / 64) * (BN / 64) * 64;
    const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
    const int lds = 2 * (BM + BN) * 32 * (int)sizeof(float);
    // (set on every launch: the attribute holds per device, a process may drive several, and a once-flag would have to be
    // per device, set only after success and thread-safe)
    RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_f32_glds_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((gemm_f32_glds_kernel<BM, BN>), dim3((unsigned)tiles, (unsigned)p.nz), dim3(THREADS), lds, s, p);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

static int launch_glds(const GemmParams& p, hipStream_t s) {
    // RSAF_GEMM_TALL=1: 256 x 128 tiles (8 waves, one workgroup per CU, 25 % less global->LDS traffic per FLOP)
    static const int tall = [] { const char* e = getenv("RSAF_GEMM_TALL"); return e ? atoi(e) : 0; }();
    if (tall && p.M >= 2048) return launch_glds_cfg<256, 128>(p, s);
    return launch_glds_cfg<128, 128>(p, s);
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
    dim3 grid((unsigned)tiles, (unsigned)p.nz);
    static const bool db = [] { const char* e = getenv("RSAF_GEMM_DB"); return e ? atoi(e) != 0 : true; }();
    if (p.b_kn) {
        if (db) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false>), grid, dim3(256), 0, s, p);
    } else {
        if (db) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, dim3(256), 0, s, p);
    }
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int launch_gemm_f32(const GemmParams& p, hipStream_t stream, const char* tag) {
    RSAF_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0 && p.nz >= 0, "negative dimension");
    if (p.M == 0 || p.N == 0 || p.nz == 0) return RSAF_OK;
    RSAF_CHECK_ARG(p.A && p.B && p.C, "NULL operand");
    RSAF_CHECK_ARG(p.nz2 >= 1 && p.nz % p.nz2 == 0, "nz must be a multiple of nz2");
    RSAF_CHECK_ARG(p.nz <= 65535, "at most 65535 batches per launch");
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    RSAF_CHECK_ARG(al16(p.A) && al16(p.B), "A and B must be 16-byte aligned");
    RSAF_CHECK_ARG(p.lda % 4 == 0 && p.ldb % 4 == 0, "lda/ldb must be multiples of 4 floats");
    RSAF_CHECK_ARG(p.sA1 % 4 == 0 && p.sA2 % 4 == 0 && p.sB1 % 4 == 0 && p.sB2 % 4 == 0,
                   "batch strides of A/B must be multiples of 4 floats");
    if (!p.b_kn) RSAF_CHECK_ARG(p.K % 4 == 0, "K must be a multiple of 4 for B[N,K]");
    else RSAF_CHECK_ARG(p.N % 4 == 0, "N must be a multiple of 4 for B[K,N]");
    RSAF_CHECK_ARG(p.a_pad_k % 4 == 0 && p.a_pad_k >= 0, "a_pad_k must be a multiple of 4");
    RSAF_CHECK_ARG(!p.R || p.ldr > 0, "residual needs ldr");
    const double flops = 2.0 * p.M * (double)p.N * p.K * p.nz;
    ProfScope prof(tag ? tag : "gemm_f32", stream, flops, 0.0);
    static const int glds = [] { const char* e = getenv("RSAF_GEMM_GLDS"); return e ? atoi(e) : 1; }();   // 0: register-staged kernel everywhere
    if (glds && !p.b_kn && p.a_pad_k == 0 && p.K % 32 == 0 && p.K >= 32 && p.N > 64) return launch_glds(p, stream);
    if (p.N <= 32) return launch_cfg<128, 32, 32, 32>(p, stream);
    if (p.N <= 64) return launch_cfg<128, 64, 64, 32>(p, stream);
    return launch_cfg<128, 128, 64, 64>(p, stream);
}

}  // namespace rsaf

using namespace rsaf;

extern "C" int rsaf_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* R,
                             int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr,
                             int nz, int nz2, const int64_t* strides8_host, int a_pad_k, int act,
                             float alpha, int b_kn, rsaf_stream_t stream) {
    GemmParams p = gemm_params_plain(A, B, C, M, N, K, lda, ldb, ldc);
    p.bias = bias; p.R = R; p.ldr = ldr; p.nz = nz; p.nz2 = nz2 < 1 ? 1 : nz2;
    if (strides8_host) {
        p.sA1 = strides8_host[0]; p.sA2 = strides8_host[1]; p.sB1 = strides8_host[2]; p.sB2 = strides8_host[3];
        p.sC1 = strides8_host[4]; p.sC2 = strides8_host[5]; p.sR1 = strides8_host[6]; p.sR2 = strides8_host[7];
    }
    RSAF_CHECK_ARG(act >= 0 && act <= 2, "act must be 0 (none), 1 (gelu) or 2 (silu)");
    p.a_pad_k = a_pad_k; p.act = act; p.alpha = alpha; p.b_kn = b_kn;
    return launch_gemm_f32(p, (hipStream_t)stream, "gemm_f32");
}
