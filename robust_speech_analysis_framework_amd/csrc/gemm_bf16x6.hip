// fp32-accurate GEMM on the bf16 matrix pipe of gfx950 (CDNA4), the dense contraction of the Wav2Vec2 stage.
//
// Replaces the stock fp32 nn.Linear / nn.Conv1d arithmetic inside transformers' Wav2Vec2Model
// (src/foundation_model_extractor.py:115 of the reference).  The fp32 matrix instruction (v_mfma_f32_32x32x2_f32)
// runs at the fp32 VECTOR rate on this chip (1/16 of the bf16 MFMA rate), and the fp32 GEMM built on it already sits
// at 86 % of the clock-adjusted peak: the only way to go faster is to spend fewer matrix cycles.
// Every fp32 operand is split into three bf16 terms, a = a1 + a2 + a3 (each the bf16 rounding of what the previous
// terms left over: 3 x 8 significand bits = the 24 bits of a float), and the product is rebuilt from the six partial
// products whose weight is at least 2^-16 of the leading one,
//     a*b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (smallest terms first).  The dropped terms are below 2^-24 |a b|
// each: the result differs from an fp32 FMA chain by a few units of fp32 rounding (checked against float64 in
// tests/test_gemm_gpu.py: the error is the same as, or smaller than, rsaf_gemm_f32's) - this is NOT a reduced-
// precision path - while the matrix pipe spends 6 x 32 cycles per 32x32x16 block instead of 8 x 64.
//
// Both operands arrive pre-split (weights once per forward call, activations by the producer's epilogue or by
// split_bf16x3_kernel).  256 x 256 x 16 block tile, 8 waves of 128 x 64 (4 x 2 MFMA tiles, 48 MFMAs per k-tile and
// wave), one workgroup per CU.  The tile size is what the data path needs: a 128 x 128 tile wants ~62 B/clk/CU from
// L2 to keep the six-product pipe busy (the chip's L2 delivers ~34.5 TB/s = 56 B/clk/CU), 256 x 256 wants 16.
// (Measured and dropped: an L2 prefetch of the lines 8 k-tiles ahead by two extra DMA instructions per k-tile into a scratch
// patch: 185 -> 151 TFLOP/s-equivalent; the fills are not what the waves wait for, the extra DMA issue slots cost more.)
// k-tiles travel global -> LDS by DMA (global_load_lds, 16 B per lane, no VGPR staging), three stages of 48 KB, two
// k-tiles in flight; the two 16-byte chunks of a 32-byte row are swapped on the SOURCE side for rows with
// (row >> 3) & 1 so that the ds_read_b128 fragment reads are conflict-free for the instruction's real lane groups
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} per half: MI355X_MICROARCH.md, LDS; (row >> 2) & 1 was 2-way).
#include <algorithm>
#include <cstdlib>

#include "gemm_bf16x6.h"

namespace rsaf {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef __attribute__((address_space(3))) void* lds_ptr6;
typedef const __attribute__((address_space(1))) void* glb_ptr6;

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

__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float4* __restrict__ src, int64_t n4,
                                                           unsigned short* __restrict__ planes, int64_t plane_stride) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = src[i];
        unsigned short h[4], m[4], l[4];
        split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]);
        split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
        *reinterpret_cast<uint2*>(planes + 4 * i) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
        *reinterpret_cast<uint2*>(planes + plane_stride + 4 * i) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
        *reinterpret_cast<uint2*>(planes + 2 * plane_stride + 4 * i) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
    }
}

// Tile geometry.  TM x TN MFMA tiles (32 x 32) per wave, WM x WN waves: block tile (32 TM WM) x (32 TN WN), k-tile 16.
//   <4,2,2,4,3,1>: 256 x 256, 3 stages of 48 KB (two k-tiles in flight), one workgroup per CU: the configuration in use.
//   (<2,2,4,2,2,2>: 256 x 128, 2 stages, two workgroups per CU, was measured: 148 / 131 / 142 / 81 TFLOP/s-equivalent on the
//    qkv / out-proj / ffn1 / ffn2 shapes against 185 / 162 / 171 / 189 for 256 x 256.)
template <int TM_, int TN_, int WM_, int WN_, int NST_, int WGS_>
struct G6Cfg {
    static constexpr int TM = TM_, TN = TN_, WM = WM_, WN = WN_, NST = NST_, WGS = WGS_;
    static constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, BK = 16;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int A_PLANE = BM * BK, B_PLANE = BN * BK;          // bf16 elements per plane per stage
    static constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
    static constexpr int A_INSTR = BM / 32, B_INSTR = BN / 32;          // 1 KiB DMA wave-instructions per plane
    static constexpr int LDS_BYTES = NST * STAGE * 2;
};

// DMA source address = wave-uniform base (kept in an SGPR pair: the empty asm stops the compiler from folding it into a
// per-lane 64-bit pointer that it then advances with vector adds) + 32-bit per-lane byte offset: the instruction's
// "saddr + voffset" form, no vector ALU work per DMA.
// (A macro, not a function: the host pass drops the kernel's stub when the kernel calls a function of this kind.)
#if defined(__HIP_DEVICE_COMPILE__)                      // register-class constraints mean nothing to the host pass
#define G6_OPAQUE(SP, LB) asm volatile("" : "+s"(SP), "+v"(LB))   // (the zero-extension has to stay next to the add to be matched)
#else
#define G6_OPAQUE(SP, LB) (void)0
#endif
#define G6_ADDR(UNIFORM_BASE, LANE_BYTES)                                                                      \
    ({                                                                                                         \
        const char* sp_ = reinterpret_cast<const char*>(UNIFORM_BASE);                                         \
        unsigned lb_ = (LANE_BYTES);                                                                           \
        G6_OPAQUE(sp_, lb_);                                                                                   \
        (glb_ptr6)(sp_ + lb_);                                                                                 \
    })

template <class CFG, int ACT, bool OUT_F32, bool OUT_PLANES, bool HAS_R>
__global__ __launch_bounds__(CFG::THREADS, CFG::WGS * CFG::THREADS / 256) void gemm_bf16x6_kernel(const Gemm6Params p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned short smem6[];
    constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK, STAGE = CFG::STAGE, NST = CFG::NST;
    constexpr int TM = CFG::TM, TN = CFG::TN, NW = CFG::WM * CFG::WN;
    constexpr int A_PLANE = CFG::A_PLANE, B_PLANE = CFG::B_PLANE;

    // the wave index as a scalar: everything derived from it (DMA bases, LDS destinations, the group of the stagger) stays
    // in SGPRs, and a DMA instruction is s_add / s_mov m0 / global_load_lds with no vector ALU work in front of it
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_n * tiles_m;
    const int orig = blockIdx.x;
    // XCD-aware bijective remap of the 1-D grid, then GROUP_M row-tiles x all column-tiles walked column by column
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int GROUP_M = p.group_m;
    const int per_group = GROUP_M * tiles_n;
    const int grp = wg / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_grp = wg - grp * per_group;
    const int m0 = (first_m + in_grp % gsz) * BM;
    const int n0 = (in_grp / gsz) * BN;
    const int wm0 = (wave / CFG::WN) * (32 * TM), wn0 = (wave % CFG::WN) * (32 * TN);
    const int64_t z = blockIdx.y;
    const int64_t z1 = p.nz2 > 1 ? z / p.nz2 : z, z2 = p.nz2 > 1 ? z % p.nz2 : 0;   // (window, group) of a grouped convolution

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // DMA: one wave-instruction moves 32 rows of one plane (1 KiB).  Instruction j of an operand covers rows 32 j ..;
    // the A_INSTR + B_INSTR instructions of a plane triple are dealt round-robin to the waves.
    // lane -> row 32 j + (lane >> 1), LDS chunk lane & 1 <- global chunk (lane & 1) ^ ((row >> 3) & 1)
    constexpr int NI = CFG::A_INSTR + CFG::B_INSTR;
    constexpr int IPW = (NI + NW - 1) / NW;                               // instructions (x 3 planes) per wave and k-tile
    const unsigned short* sbase[IPW];                                    // wave-uniform: operand + tile origin (SGPR pair)
    unsigned voff[IPW];                                                   // per lane: ((row in tile) * ld + chunk) * 2 bytes
    int64_t pstride[IPW], kstride[IPW];
    int ldsoff[IPW];
    bool live[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int j = wave + NW * i;
        live[i] = j < NI;
        const bool isA = j < CFG::A_INSTR;
        const int jj = isA ? j : j - CFG::A_INSTR;
        const int row = 32 * jj + (lane >> 1);
        const int dch = (lane & 1) ^ ((row >> 3) & 1);
        if (isA) {
            const int rr = (m0 + row < p.M) ? row : p.M - 1 - m0;         // rows past M re-read the last row
            const int64_t rs = p.a_panel ? 16 : p.lda;                    // row stride; k-tile stride below (elements)
            sbase[i] = p.A + z1 * p.sA + z2 * p.sA2 + (int64_t)m0 * rs;
            voff[i] = 2u * ((unsigned)rr * (unsigned)rs + 8u * dch);
            pstride[i] = p.a_plane;
            kstride[i] = p.a_panel ? (int64_t)p.M * 16 : 16;
            ldsoff[i] = jj * 512;
        } else {
            const int rr = (n0 + row < p.N) ? row : p.N - 1 - n0;
            const int64_t rs = p.b_panel ? 16 : p.ldb;
            sbase[i] = p.B + z2 * p.sB2 + (int64_t)n0 * rs;
            voff[i] = 2u * ((unsigned)rr * (unsigned)rs + 8u * dch);
            pstride[i] = p.b_plane;
            kstride[i] = p.b_panel ? (int64_t)(p.b_panel_rows > 0 ? p.b_panel_rows : p.N) * 16 : 16;
            ldsoff[i] = 3 * A_PLANE + jj * 512;
        }
    }
    const int nk = p.K / BK;

#define G6_DMA(KT, ST)                                                                                          \
    do {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < IPW; ++i) {                                                       \
            if (live[i]) {                                                                                      \
                const bool isA_ = (wave + NW * i) < CFG::A_INSTR;                                               \
                _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                \
                    __builtin_amdgcn_global_load_lds(G6_ADDR(sbase[i] + (pl * pstride[i] + (int64_t)(KT) * kstride[i]), voff[i]), \
                        (lds_ptr6)(smem6 + (ST) * STAGE + ldsoff[i] + pl * (isA_ ? A_PLANE : B_PLANE)), 16, 0, 0); \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)

    // Main loop: ONE instruction stream, two barriers per k-tile, and the two waves of every SIMD half a k-tile apart.
    // A k-tile of a wave is two halves: m-tiles 0 .. TM/2-1 (the B fragments and the A fragments come from LDS), a barrier,
    // m-tiles TM/2 .. TM-1 (their first A fragments were fetched in front of that barrier and stay in flight across it).
    // Waves NW/2 .. NW-1 (group B; they share the four SIMDs with waves 0 .. NW/2-1) pass one extra barrier in front of the
    // loop and the others one behind it: barriers match by count, so B runs the same stream one half k-tile later.  While
    // one wave of a SIMD waits for the LDS burst of a new k-tile, the other has 24 MFMAs whose operands are in registers,
    // and the 8 waves no longer read the LDS at once.  With slots numbered by barriers (A runs half s in slot s, B half s-1):
    //   * a wave issues its six DMA instructions at the HEAD of its first half, behind the B-fragment reads of the new
    //     k-tile: it waits for those reads anyway, and its SIMD partner is in a second half that needs nothing from LDS
    //     (between the MFMA groups of the second half the same instructions cost 4-8 % more; in front of the fragment
    //     reads, behind the first A-fragment reads or behind the first MFMA group 2-5 % more);
    //   * group A (slot 2 kt) fetches k-tile kt + 1 into the stage of k-tile kt - 2 (last read by B in slot 2 kt - 2) and
    //     waits for it (vmcnt 0) in front of its next top barrier, 2 kt + 2; group B (slot 2 kt + 1) fetches k-tile kt + 2
    //     into the stage of k-tile kt - 1 (last read by B itself in slot 2 kt) and waits for it in front of the middle
    //     barrier of its k-tile kt + 1 (barrier 2 kt + 4, counted: its next six DMA instructions stay in flight).  The first
    //     reader of a k-tile is always A behind its top barrier: one k-tile of cover for A's share, 1.5 for B's
    //     (two stages ran within 2 % of three: the loop is not waiting for the data).
    static_assert(NST == 3 && TM % 2 == 0, "the staggered loop needs three stages and an even number of m-tiles");
    G6_DMA(0, 0);
    if (nk > 1) G6_DMA(1, 1);
    const int grpB = wave >= NW / 2 ? 1 : 0;
    constexpr int NDMA = 3 * IPW;                            // DMA instructions of this wave per k-tile
    constexpr int HM = TM / 2;
    if (nk > 1) {                                            // k-tile 0 has landed (k-tile 1 may stay in flight)
        if (IPW == 1 || !live[IPW - 1]) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (IPW - 1) > 0 ? 3 * (IPW - 1) : 3) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * IPW) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (grpB) __builtin_amdgcn_s_barrier();
    int st = 0;                                              // stage of k-tile kt
    for (int kt = 0; kt < nk; ++kt) {
        if (!grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // A: its share of k-tile kt (issued one k-tile ago)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // DMA at the head of the first half, behind the fragment reads of the new k-tile (the wave waits for those anyway):
        // group A fetches k-tile kt + 1 (stage of kt - 2), group B - half a k-tile later in time - k-tile kt + 2 (stage of kt - 1)
        const int rkt = kt + 1 + grpB;
        const bool refill = rkt < nk && (grpB || kt >= 1);
        const int rst = (st + 1 + grpB) % NST;
        const unsigned short* img = smem6 + st * STAGE;
        bf16x8 bf[TN][3];
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int row = wn0 + nt * 32 + l31;
            const int off = row * BK + ((h ^ ((row >> 3) & 1)) << 3);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[nt][pl] = *reinterpret_cast<const bf16x8*>(&img[3 * A_PLANE + pl * B_PLANE + off]);
        }
        if (refill) {
#pragma unroll
            for (int d = 0; d < NDMA; ++d) {
                const int i = d / 3, pl = d % 3;
                if (live[i]) {
                    const bool isA_ = (wave + NW * i) < CFG::A_INSTR;
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_global_load_lds(G6_ADDR(sbase[i] + (pl * pstride[i] + (int64_t)rkt * kstride[i]), voff[i]),
                        (lds_ptr6)(smem6 + rst * STAGE + ldsoff[i] + pl * (isA_ ? A_PLANE : B_PLANE)), 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        bf16x8 afp[3];                                       // A fragments of m-tile HM, fetched in the first half
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            bf16x8 af[3];
            if (mt == HM) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[pl] = afp[pl];
            } else {
                const int row = wm0 + mt * 32 + l31;
                const int off = row * BK + ((h ^ ((row >> 3) & 1)) << 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(&img[pl * A_PLANE + off]);
            }
            if (mt == HM - 1) {
                const int row = wm0 + HM * 32 + l31;
                const int off = row * BK + ((h ^ ((row >> 3) & 1)) << 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) afp[pl] = *reinterpret_cast<const bf16x8*>(&img[pl * A_PLANE + off]);
            }
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                f32x16 c = acc[mt][nt];                      // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[nt][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[nt][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[nt][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[nt][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[nt][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[nt][0], c, 0, 0, 0);
                acc[mt][nt] = c;
            }
            if (mt == HM - 1) {
                // middle barrier: group B's share of k-tile kt + 1 (issued 1.5 k-tiles ago) has landed; its share of
                // k-tile kt + 2, issued at the head of this k-tile, stays in flight
                if (grpB) {                                  // (the count is this wave's own: the last instruction slot may be empty)
                    if (!refill) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (live[IPW - 1]) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * IPW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (IPW - 1) > 0 ? 3 * (IPW - 1) : 0) : "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        st = st == NST - 1 ? 0 : st + 1;
    }
    if (!grpB) __builtin_amdgcn_s_barrier();
#undef G6_DMA

    // ---- epilogue ----
    // MFMA layout: lane l31 = column, register e -> row (e & 3) + 8 (e >> 2) + 4 h of a 32 x 32 tile.  Storing from that
    // layout is 4 (fp32) or 2 (bf16) bytes per lane and store: 128 / 384 store instructions per lane, store-issue bound.
    // Each tile is transposed through a wave-private LDS patch (the k-tile stages are free now) into rows: lane ->
    // row lane >> 1, 16 consecutive columns, then bias / residual / activation in that layout and 16-byte stores.
    __syncthreads();                                     // every wave is done reading the last k-tile
    constexpr int PATCH_LD = 36;                         // floats per patch row (16-byte aligned rows, 2-way write conflicts)
    float* patch = reinterpret_cast<float*>(smem6) + wave * (32 * PATCH_LD);
    const int prow = lane >> 1, pcol = 16 * (lane & 1);
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
        const int tn = n0 + wn0 + nt * 32;
        const int gc = tn + pcol;                        // first of this lane's 16 columns
        const bool c_ok = gc < p.N;                      // N % 16 == 0: the 16 columns are valid together
        float bias16[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = (p.bias && c_ok) ? *reinterpret_cast<const float4*>(p.bias + z2 * p.sBias2 + gc + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            bias16[4 * q] = b4.x; bias16[4 * q + 1] = b4.y; bias16[4 * q + 2] = b4.z; bias16[4 * q + 3] = b4.w;
        }
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int tm = m0 + wm0 + mt * 32;
            const bool ok = c_ok && (tm + prow) < p.M;
            float4 r4[4];
            if (HAS_R) {
                const float* Rt = p.R + z * p.sR + (int64_t)tm * p.ldr + tn;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r4[q] = ok ? *reinterpret_cast<const float4*>(Rt + prow * (int)p.ldr + pcol + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) patch[(4 * h + (e & 3) + 8 * (e >> 2)) * PATCH_LD + l31] = p.alpha * acc[mt][nt][e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t4 = *reinterpret_cast<const float4*>(patch + prow * PATCH_LD + pcol + 4 * q);
                v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
            }
            __builtin_amdgcn_wave_barrier();             // the patch may be overwritten by the next tile
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float x = v[q] + bias16[q];
                if (HAS_R) x += (q & 3) == 0 ? r4[q >> 2].x : ((q & 3) == 1 ? r4[q >> 2].y : ((q & 3) == 2 ? r4[q >> 2].z : r4[q >> 2].w));
                if (ACT == ACT_GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
                if (ACT == ACT_SILU) x = x / (1.0f + expf(-x));
                v[q] = x;
            }
            if (ok) {
                if (OUT_F32) {
                    float* Ct = p.C + z1 * p.sC + z2 * p.sC2 + (int64_t)tm * p.ldc + tn + prow * (int)p.ldc + pcol;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(Ct + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                }
                if (OUT_PLANES) {
                    unsigned short hh[16], mm[16], ll[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) split3(v[q], hh[q], mm[q], ll[q]);
                    // row-major [M][ldcp], or the k16 panels of the next GEMM's A: panel (tn + pcol) / 16, row tm + prow
                    unsigned short* Pt = p.cp_panel ? p.Cp + (int64_t)((tn + pcol) >> 4) * ((int64_t)p.M * 16) + (int64_t)(tm + prow) * 16
                                                    : p.Cp + z * p.sCp + (int64_t)tm * p.ldcp + tn + prow * (int)p.ldcp + pcol;
                    auto pk = [](const unsigned short* s_, int o) {
                        return make_uint4(s_[o] | ((unsigned)s_[o + 1] << 16), s_[o + 2] | ((unsigned)s_[o + 3] << 16),
                                          s_[o + 4] | ((unsigned)s_[o + 5] << 16), s_[o + 6] | ((unsigned)s_[o + 7] << 16));
                    };
                    *reinterpret_cast<uint4*>(Pt) = pk(hh, 0);
                    *reinterpret_cast<uint4*>(Pt + 8) = pk(hh, 8);
                    *reinterpret_cast<uint4*>(Pt + p.c_plane) = pk(mm, 0);
                    *reinterpret_cast<uint4*>(Pt + p.c_plane + 8) = pk(mm, 8);
                    *reinterpret_cast<uint4*>(Pt + 2 * p.c_plane) = pk(ll, 0);
                    *reinterpret_cast<uint4*>(Pt + 2 * p.c_plane + 8) = pk(ll, 8);
                }
            }
        }
    }
}

// fp32 [rows][K] -> planes in k16 panels: a thread takes 16 consecutive k of one row (64 bytes in, 32 bytes per plane out),
// lanes take consecutive rows, so a wave writes 2 KiB contiguous per plane
__global__ __launch_bounds__(256) void split_bf16x3_panels_kernel(const float* __restrict__ src, int64_t rows, int K,
                                                                  unsigned short* __restrict__ planes, int64_t plane_stride) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int kb = blockIdx.y;                                       // panel
    const float4* s4 = reinterpret_cast<const float4*>(src + row * K + 16 * kb);
    unsigned short h[16], m[16], l[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = s4[q];
        split3(v.x, h[4 * q], m[4 * q], l[4 * q]); split3(v.y, h[4 * q + 1], m[4 * q + 1], l[4 * q + 1]);
        split3(v.z, h[4 * q + 2], m[4 * q + 2], l[4 * q + 2]); split3(v.w, h[4 * q + 3], m[4 * q + 3], l[4 * q + 3]);
    }
    auto pk = [](const unsigned short* s_, int o) {
        return make_uint4(s_[o] | ((unsigned)s_[o + 1] << 16), s_[o + 2] | ((unsigned)s_[o + 3] << 16),
                          s_[o + 4] | ((unsigned)s_[o + 5] << 16), s_[o + 6] | ((unsigned)s_[o + 7] << 16));
    };
    unsigned short* d = planes + (int64_t)kb * rows * 16 + row * 16;
    *reinterpret_cast<uint4*>(d) = pk(h, 0); *reinterpret_cast<uint4*>(d + 8) = pk(h, 8);
    *reinterpret_cast<uint4*>(d + plane_stride) = pk(m, 0); *reinterpret_cast<uint4*>(d + plane_stride + 8) = pk(m, 8);
    *reinterpret_cast<uint4*>(d + 2 * plane_stride) = pk(l, 0); *reinterpret_cast<uint4*>(d + 2 * plane_stride + 8) = pk(l, 8);
}

int launch_split_bf16x3_panels(const float* src, int64_t rows, int K, uint16_t* planes, int64_t plane_stride, hipStream_t stream) {
    RSAF_CHECK_ARG(rows >= 0 && K >= 0 && K % 16 == 0, "K must be a non-negative multiple of 16");
    if (rows == 0 || K == 0) return RSAF_OK;
    RSAF_CHECK_ARG(src && planes, "NULL pointer");
    RSAF_CHECK_ARG(plane_stride % 8 == 0 && K / 16 <= 65535, "plane stride must be a multiple of 8 elements; at most 65535 panels");
    ProfScope prof("split_bf16x3", stream, 0.0, 10.0 * (double)rows * K);
    hipLaunchKernelGGL(split_bf16x3_panels_kernel, dim3((unsigned)((rows + 255) / 256), (unsigned)(K / 16)), dim3(256), 0, stream, src,
                       rows, K, reinterpret_cast<unsigned short*>(planes), plane_stride);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int launch_split_bf16x3(const float* src, int64_t n, uint16_t* planes, int64_t plane_stride, hipStream_t stream) {
    RSAF_CHECK_ARG(n >= 0 && n % 4 == 0, "length must be a non-negative multiple of 4");
    if (n == 0) return RSAF_OK;
    RSAF_CHECK_ARG(src && planes, "NULL pointer");
    RSAF_CHECK_ARG(plane_stride % 4 == 0, "plane stride must be a multiple of 4 elements");
    const int64_t n4 = n / 4;
    const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 256 * 64);
    ProfScope prof("split_bf16x3", stream, 0.0, 10.0 * (double)n);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(src), n4,
                       reinterpret_cast<unsigned short*>(planes), plane_stride);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int launch_gemm_bf16x6(const Gemm6Params& p, hipStream_t stream, const char* tag) {
    RSAF_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0 && p.nz >= 0, "negative dimension");
    if (p.M == 0 || p.N == 0 || p.nz == 0) return RSAF_OK;
    RSAF_CHECK_ARG(p.A && p.B && (p.C || p.Cp), "NULL operand");
    RSAF_CHECK_ARG(p.K % 16 == 0 && p.K >= 16, "K must be a positive multiple of 16");
    RSAF_CHECK_ARG(p.N % 16 == 0, "N must be a multiple of 16 (16-byte epilogue stores)");
    RSAF_CHECK_ARG(p.ldc % 8 == 0 && p.ldcp % 8 == 0 && p.ldr % 4 == 0 && p.c_plane % 8 == 0 && p.sC % 4 == 0 && p.sCp % 8 == 0 &&
                   p.sR % 4 == 0, "ldc / ldr / output strides must keep 16-byte alignment");
    RSAF_CHECK_ARG(p.lda % 8 == 0 && p.ldb % 8 == 0 && p.a_plane % 8 == 0 && p.b_plane % 8 == 0 && p.sA % 8 == 0,
                   "lda, ldb, the plane strides and the batch stride of A must be multiples of 8 elements");
    RSAF_CHECK_ARG((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0,
                   "A and B must be 16-byte aligned");
    RSAF_CHECK_ARG(!p.R || p.ldr > 0, "residual needs ldr");
    RSAF_CHECK_ARG(p.act >= 0 && p.act <= 2, "act must be 0 (none), 1 (gelu) or 2 (silu)");
    RSAF_CHECK_ARG(p.nz <= 65535, "at most 65535 batches per launch");
    RSAF_CHECK_ARG(!(p.a_panel || p.cp_panel) || p.nz == 1, "panel layouts are for unbatched operands");
    RSAF_CHECK_ARG(p.ldc < (1 << 24) && p.ldcp < (1 << 24) && p.ldr < (1 << 24), "leading dimensions must be below 2^24");
    // algorithmic FLOPs of the contraction (2 M N K); the matrix pipe executes six bf16 products per term
    ProfScope prof(tag ? tag : "gemm_bf16x6", stream, 2.0 * p.M * (double)p.N * p.K * p.nz, 0.0);
    using CfgA = G6Cfg<4, 2, 2, 4, 3, 1>;                // 256 x 256 (a 256 x 128 / two-workgroup variant measured 10-50 % slower)
    using CfgN = G6Cfg<2, 1, 4, 2, 3, 1>;                // 256 x 64: N <= 64 (the 48-wide groups of the positional convolution)
    RSAF_CHECK_ARG(p.nz2 <= 1 || (p.C && !p.Cp && !p.R && !p.a_panel && p.nz % p.nz2 == 0),
                   "two-level batches: fp32 output only, no residual, A row-major, nz a multiple of nz2");
#define G6_LAUNCH_CFG(CFG, ACT, F32, PL, HR)                                                                            \
    do {                                                                                                                \
        static DeviceOnce attr_once;                                                                                    \
        if (attr_once.first()) {                                                                                        \
            RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16x6_kernel<CFG, ACT, F32, PL, HR>,                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES));             \
        }                                                                                                               \
        const int64_t tiles = (int64_t)((p.N + CFG::BN - 1) / CFG::BN) * ((p.M + CFG::BM - 1) / CFG::BM);               \
        RSAF_CHECK_ARG(tiles <= 0x7fffffffLL, "too many tiles");                                                        \
        hipLaunchKernelGGL((gemm_bf16x6_kernel<CFG, ACT, F32, PL, HR>), dim3((unsigned)tiles, (unsigned)p.nz),          \
                           dim3(CFG::THREADS), CFG::LDS_BYTES, stream, pp);                                              \
    } while (0)
#define G6_LAUNCH(ACT, F32, PL, HR) G6_LAUNCH_CFG(CfgA, ACT, F32, PL, HR)
    Gemm6Params pp = p;
    if (pp.group_m <= 0) pp.group_m = 2;                 // measured: 1, 2, 4, 8 within 2 %, 16 is 5 % slower
    const bool f32o = p.C != nullptr, plo = p.Cp != nullptr, hr = p.R != nullptr;
    // the combinations the Wav2Vec2 / CNN stages use (anything else is an argument error, not a silent fallback)
    if (p.act == ACT_NONE && f32o && !plo && !hr) G6_LAUNCH(ACT_NONE, true, false, false);
    else if (p.act == ACT_NONE && f32o && !plo && hr) G6_LAUNCH(ACT_NONE, true, false, true);
    else if (p.act == ACT_GELU && !f32o && plo && !hr) G6_LAUNCH(ACT_GELU, false, true, false);
    else if (p.act == ACT_GELU && f32o && !plo && !hr && p.N <= 64) G6_LAUNCH_CFG(CfgN, ACT_GELU, true, false, false);
    else if (p.act == ACT_GELU && f32o && !plo && !hr) G6_LAUNCH(ACT_GELU, true, false, false);
    else if (p.act == ACT_NONE && !f32o && plo && !hr) G6_LAUNCH(ACT_NONE, false, true, false);
    else if (p.act == ACT_NONE && f32o && plo && !hr) G6_LAUNCH(ACT_NONE, true, true, false);
    else {
        set_error("launch_gemm_bf16x6: unsupported combination of activation / outputs / residual");
        return RSAF_ERR_ARG;
    }
#undef G6_LAUNCH_CFG
#undef G6_LAUNCH
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

}  // namespace rsaf

using namespace rsaf;

extern "C" int rsaf_split_bf16x3(const float* src, int64_t n, uint16_t* planes, int64_t plane_stride, rsaf_stream_t stream) {
    return launch_split_bf16x3(src, n, planes, plane_stride, (hipStream_t)stream);
}

extern "C" int rsaf_gemm_bf16x6(const uint16_t* A_planes, int64_t a_plane_stride, const uint16_t* B_planes,
                                int64_t b_plane_stride, float* C, uint16_t* C_planes, int64_t c_plane_stride,
                                const float* bias, const float* R, int M, int N, int K, int64_t lda, int64_t ldb,
                                int64_t ldc, int64_t ldr, int act, float alpha, rsaf_stream_t stream) {
    Gemm6Params p{};
    p.A = A_planes; p.a_plane = a_plane_stride; p.lda = lda; p.sA = 0;
    p.B = B_planes; p.b_plane = b_plane_stride; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.sC = 0;
    p.Cp = C_planes; p.c_plane = c_plane_stride; p.ldcp = ldc; p.sCp = 0;
    p.bias = bias; p.R = R; p.ldr = ldr; p.sR = 0;
    p.M = M; p.N = N; p.K = K; p.nz = 1; p.act = act; p.alpha = alpha;
    return launch_gemm_bf16x6(p, (hipStream_t)stream, "gemm_bf16x6");
}

extern "C" int rsaf_split_bf16x3_panels(const float* src, int64_t rows, int K, uint16_t* planes, int64_t plane_stride,
                                        rsaf_stream_t stream) {
    return launch_split_bf16x3_panels(src, rows, K, planes, plane_stride, (hipStream_t)stream);
}

extern "C" int rsaf_gemm_bf16x6_panels(const uint16_t* A_planes, int64_t a_plane_stride, const uint16_t* B_planes,
                                       int64_t b_plane_stride, float* C, uint16_t* C_planes, int64_t c_plane_stride,
                                       const float* bias, const float* R, int M, int N, int K, int64_t lda, int64_t ldb,
                                       int64_t ldc, int64_t ldr, int act, float alpha, int a_panels, int b_panels,
                                       int c_panels, rsaf_stream_t stream) {
    Gemm6Params p{};
    p.A = A_planes; p.a_plane = a_plane_stride; p.lda = a_panels ? 16 : lda; p.sA = 0;
    p.B = B_planes; p.b_plane = b_plane_stride; p.ldb = b_panels ? 16 : ldb;
    p.C = C; p.ldc = ldc; p.sC = 0;
    p.Cp = C_planes; p.c_plane = c_plane_stride; p.ldcp = c_panels ? 16 : ldc; p.sCp = 0;
    p.bias = bias; p.R = R; p.ldr = ldr; p.sR = 0;
    p.M = M; p.N = N; p.K = K; p.nz = 1; p.act = act; p.alpha = alpha;
    p.a_panel = a_panels != 0; p.b_panel = b_panels != 0; p.cp_panel = c_panels != 0;
    return launch_gemm_bf16x6(p, (hipStream_t)stream, "gemm_bf16x6");
}
