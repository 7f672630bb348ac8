// fp32-accurate GEMM on the fp16 matrix pipe of gfx950 (CDNA4): the dense contraction of the Wav2Vec2 stage.
//
// Replaces the stock fp32 nn.Linear / nn.Conv1d arithmetic inside transformers' Wav2Vec2Model
// (src/foundation_model_extractor.py:115 of the reference).  The fp32 matrix instruction (v_mfma_f32_32x32x2_f32)
// runs at the fp32 VECTOR rate on this chip (1/16 of the 16-bit MFMA rate), so an fp32 result has to be rebuilt from
// 16-bit products.  Rounds 2-3 used three bf16 terms per operand and six products (gemm_bf16x6); this kernel gets the
// same accuracy from THREE products:
//     every operand row is first multiplied by a power of two s that puts its largest magnitude just below 2^15
//     (exact), then split into two fp16 terms  x s = h + l,  h = fp16(x s),  l = fp16(x s - h):
//     11 + 11 significand bits, |x s - h - l| <= 2^-22 |x s|, and as long as |x s| >= 2^-3 both terms are normal fp16
//     numbers (below that the absolute error stays under 2^-25, i.e. 2^-40 of the row's maximum: graceful);
//     a b = (a_h b_h + (a_h b_l + a_l b_h)) / (s_a s_b)   with the dropped a_l b_l <= 2^-22 |a b|,
// accumulated in fp32 by v_mfma_f32_32x32x16_f16 (small terms first), one rounding per 16 products.  Against float64
// the error is that of (or below) an fp32 FMA chain, which rounds K times (tests/test_gemm_gpu.py asserts it against
// rsaf_gemm_f32's own error on the same operands, with tight and with 4 096 x loose scales): NOT a reduced-precision
// path.  Operand bytes are 4 per element (6 with three bf16 planes), matrix cycles half.
//
// Scales: weights get their exact row maximum at split time; activations get it from their producer - LayerNorm knows
// its row, a GEMM epilogue cannot know the maximum of the row it is still computing and uses a Cauchy-Schwarz bound
// (|x_mn| <= |a_m|_2 |w_n|_2 + |b_n|) handed in as `c_scale`; every epilogue can also report the true maximum of what it
// wrote (`amax_out`) for the next producer's bound.
//
// Structure (unchanged from the bf16x6 kernel it replaces; DESIGN.md 4.1 has the measurements behind each choice):
// 256 x 256 x 16 block tile, 8 waves of 128 x 64 (4 x 2 MFMA tiles, 24 MFMAs per k-tile and wave), one workgroup per
// CU; k-tiles travel global -> LDS by DMA (global_load_lds, 16 B per lane), three stages of 32 KB, two k-tiles in
// flight; the two 16-byte chunks of a 32-byte row are swapped on the SOURCE side for rows with (row >> 3) & 1 so that
// the ds_read_b128 fragment reads are conflict-free; the two waves of a SIMD run half a k-tile apart.
#include <algorithm>
#include <cstdlib>

#include "gemm_f16x3.h"

namespace rsaf {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((address_space(3))) void* lds_ptr3;
typedef const __attribute__((address_space(1))) void* glb_ptr3;

__device__ __forceinline__ unsigned short f16_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }

// xs = h + l (+ at most 2^-22 |xs|); xs already carries the row's scale
__device__ __forceinline__ void split2(float xs, unsigned short& h, unsigned short& l) {
    const _Float16 hh = (_Float16)xs;
    const _Float16 ll = (_Float16)(xs - (float)hh);
    h = f16_bits(hh);
    l = f16_bits(ll);
}

template <int TM_, int TN_, int WM_, int WN_, int NST_, int WGS_>
struct H3Cfg {
    static constexpr int TM = TM_, TN = TN_, WM = WM_, WN = WN_, NST = NST_, WGS = WGS_;
    static constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, BK = 16, NPL = 2;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int A_PLANE = BM * BK, B_PLANE = BN * BK;          // fp16 elements per plane per stage
    static constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
    static constexpr int A_INSTR = BM / 32, B_INSTR = BN / 32;          // 1 KiB DMA wave-instructions per plane
    static constexpr int PATCH_BYTES = WM * WN * 32 * 36 * 4;           // wave-private transposition patches of the epilogue
    static constexpr int TAB_BYTES = 2 * (BM + BN) * 4;                 // the tile's row / column scales and biases (epilogue)
    // The patches do not alias stages 0 and 1: the next tile's first two k-tiles land there while the epilogue runs.  Stage 2
    // is free from the end of a tile's main loop until the next tile's first refill (issued inside that tile's main loop):
    // the 512 x 128 configuration, whose three 40 KB stages leave no room beside them, keeps its patches there.
    static constexpr bool PATCH_IN_STAGE2 = BM > 256;
    static_assert(!PATCH_IN_STAGE2 || STAGE * 2 >= PATCH_BYTES, "stage 2 must hold the epilogue patches");
    static constexpr int LDS_BYTES = NST * STAGE * 2 + (PATCH_IN_STAGE2 ? 0 : PATCH_BYTES) + TAB_BYTES;
};

// DMA source address = wave-uniform base (kept in an SGPR pair) + 32-bit per-lane byte offset: the instruction's
// "saddr + voffset" form, no vector ALU work per DMA (see gemm_bf16x6 history in DESIGN.md 4.1).
#if defined(__HIP_DEVICE_COMPILE__)
#define H3_OPAQUE(SP, LB) asm volatile("" : "+s"(SP), "+v"(LB))
#else
#define H3_OPAQUE(SP, LB) (void)0
#endif
#define H3_ADDR(UNIFORM_BASE, LANE_BYTES)                                                                      \
    ({                                                                                                         \
        const char* sp_ = reinterpret_cast<const char*>(UNIFORM_BASE);                                         \
        unsigned lb_ = (LANE_BYTES);                                                                           \
        H3_OPAQUE(sp_, lb_);                                                                                   \
        (glb_ptr3)(sp_ + lb_);                                                                                 \
    })

template <class CFG, int ACT, bool OUT_F32, bool OUT_PLANES, bool HAS_R>
__global__ __launch_bounds__(CFG::THREADS, CFG::WGS * CFG::THREADS / 256) void gemm_f16x3_kernel(const GemmH3Params p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned short smem3[];
    constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK, STAGE = CFG::STAGE, NST = CFG::NST, NPL = CFG::NPL;
    constexpr int TM = CFG::TM, TN = CFG::TN, NW = CFG::WM * CFG::WN;
    constexpr int A_PLANE = CFG::A_PLANE, B_PLANE = CFG::B_PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_n * tiles_m;
    const int64_t total_tiles = (int64_t)nwg * p.nz;
    const int wm0 = (wave / CFG::WN) * (32 * TM), wn0 = (wave % CFG::WN) * (32 * TN);

    // Persistent workgroups: workgroup b walks tiles b, b + gridDim.x, ...  Tile t -> batch z = t / nwg and, inside the batch,
    // the XCD-aware bijective remap of the index, then GROUP_M row-tiles x all column-tiles walked column by column.
    struct Tile { int m0, n0, Mz; int64_t z, z1, z2, coff; };
    auto tile_at = [&](int64_t t) {
        Tile T;
        T.z = t / nwg;
        const int orig = (int)(t - T.z * nwg);
        T.z1 = p.nz2 > 1 ? T.z / p.nz2 : T.z;
        T.z2 = p.nz2 > 1 ? T.z % p.nz2 : 0;                              // (window, group) of a grouped convolution
        T.Mz = p.ztab ? (int)p.ztab[2 * T.z1] : p.M;                     // rows of this batch (ragged windows)
        T.coff = p.ztab ? p.ztab[2 * T.z1 + 1] : -1;
        const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
        const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
        const int GROUP_M = p.group_m;
        const int per_group = GROUP_M * tiles_n;
        const int grp = wg / per_group;
        const int first_m = grp * GROUP_M;
        const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
        const int in_grp = wg - grp * per_group;
        T.m0 = (first_m + in_grp % gsz) * BM;
        T.n0 = (in_grp / gsz) * BN;
        return T;
    };
    // the first tile at or after t (stride gridDim.x) that has rows: tiles past a short batch's rows are skipped
    auto next_valid = [&](int64_t t, Tile& T) {
        for (; t < total_tiles; t += gridDim.x) {
            T = tile_at(t);
            if (T.m0 < T.Mz) return t;
        }
        return (int64_t)-1;
    };

    // DMA: one wave-instruction moves 32 rows of one plane (1 KiB).  Instruction j of an operand covers rows 32 j ..;
    // the A_INSTR + B_INSTR instructions of a plane pair are dealt round-robin to the waves.
    // lane -> row 32 j + (lane >> 1), LDS chunk lane & 1 <- global chunk (lane & 1) ^ ((row >> 3) & 1)
    constexpr int NI = CFG::A_INSTR + CFG::B_INSTR;
    constexpr int IPW = (NI + NW - 1) / NW;                               // instructions (x NPL planes) per wave and k-tile
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
        pstride[i] = isA ? p.a_plane : p.b_plane;
        kstride[i] = isA ? (p.a_panel ? (int64_t)(p.a_panel_rows > 0 ? p.a_panel_rows : p.M) * 16 : 16) : (p.b_panel ? (int64_t)(p.b_panel_rows > 0 ? p.b_panel_rows : p.N) * 16 : 16);
        ldsoff[i] = isA ? jj * 512 : NPL * A_PLANE + jj * 512;
    }
    auto set_tile = [&](const Tile& T) {                                  // DMA sources of a tile
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int j = wave + NW * i;
            const bool isA = j < CFG::A_INSTR;
            const int jj = isA ? j : j - CFG::A_INSTR;
            const int row = 32 * jj + (lane >> 1);
            const int dch = (lane & 1) ^ ((row >> 3) & 1);
            if (isA) {
                const int rr = (T.m0 + row < T.Mz) ? row : T.Mz - 1 - T.m0;   // rows past M re-read the last row
                const int64_t rs = p.a_panel ? 16 : p.lda;                // row stride (elements)
                sbase[i] = p.A + T.z1 * p.sA + T.z2 * p.sA2 + (int64_t)T.m0 * rs;
                voff[i] = 2u * ((unsigned)rr * (unsigned)rs + 8u * dch);
            } else {
                const int rr = (T.n0 + row < p.N) ? row : p.N - 1 - T.n0;
                const int64_t rs = p.b_panel ? 16 : p.ldb;
                sbase[i] = p.B + T.z2 * p.sB2 + (int64_t)T.n0 * rs;
                voff[i] = 2u * ((unsigned)rr * (unsigned)rs + 8u * dch);
            }
        }
    };
    const int nk = p.K / BK;
    // element offset of k-tile KT of instruction i's operand.  A convolution over a padded panel image (a_tap_panels = P > 0:
    // K = taps x P panels) walks the P channel panels of a tap and then moves ONE ROW down for the next tap: the rows of
    // consecutive taps overlap, the image holds every activation once.  tap = KT / P by a host-checked reciprocal.
    auto koff = [&](int i, int KT) -> int64_t {
        if (p.a_tap_panels > 0 && (wave + NW * i) < CFG::A_INSTR) {
            const int tap = (int)(((unsigned)KT * (unsigned)p.a_tap_inv) >> 16);
            return (int64_t)(KT - tap * p.a_tap_panels) * kstride[i] + (int64_t)tap * 16;
        }
        return (int64_t)KT * kstride[i];
    };

#define H3_DMA(KT, ST)                                                                                          \
    do {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < IPW; ++i) {                                                       \
            if (live[i]) {                                                                                      \
                const bool isA_ = (wave + NW * i) < CFG::A_INSTR;                                               \
                _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                              \
                    __builtin_amdgcn_global_load_lds(H3_ADDR(sbase[i] + (pl * pstride[i] + koff(i, (KT))), voff[i]), \
                        (lds_ptr3)(smem3 + (ST) * STAGE + ldsoff[i] + pl * (isA_ ? A_PLANE : B_PLANE)), 16, 0, 0); \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)

    static_assert(NST == 3 && TM % 2 == 0, "the staggered loop needs three stages and an even number of m-tiles");
    const int grpB = wave >= NW / 2 ? 1 : 0;
    constexpr int NDMA = NPL * IPW;                          // DMA instructions of this wave per k-tile
    constexpr int HM = TM / 2;
    constexpr int PATCH_LD = 36;                             // floats per patch row (16-byte aligned rows, 2-way write conflicts)
    float* patch = reinterpret_cast<float*>(smem3 + (CFG::PATCH_IN_STAGE2 ? 2 : NST) * STAGE) + wave * (32 * PATCH_LD);

    // The epilogue's per-row and per-column operands (scales of A and of the plane output; scales of B and bias) reach it
    // through LDS: thread t fetches row t's (t < BM) or column t - BM's pair when the tile STARTS, keeps it in two registers
    // through the main loop and files it behind the last k-tile.  (As global loads inside the epilogue they sat behind the
    // stores in the wave's in-order memory counter, or cost 30+ registers when all fetched up front.)
    constexpr int TPT = (BM + BN + CFG::THREADS - 1) / CFG::THREADS;     // table entries per thread (1; 2 for the 512 x 128 tile)
    float* tab0 = reinterpret_cast<float*>(smem3 + NST * STAGE) + (CFG::PATCH_IN_STAGE2 ? 0 : NW * (32 * PATCH_LD));   // [BM] a_scale | [BN] b_scale
    float* tab1 = tab0 + (BM + BN);                                                        // [BM] c_scale | [BN] bias
    float e0[TPT], e1[TPT];
    auto load_epi = [&](const Tile& T) {
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int i = tid + u * CFG::THREADS;
            e0[u] = 1.0f; e1[u] = 0.0f;
            if (i < BM) {
                const int row = T.m0 + i < T.Mz ? T.m0 + i : T.Mz - 1;                      // clamped: rows past M are never stored
                e0[u] = p.a_scale[T.z1 * p.a_scale_zs + (int64_t)row * p.a_scale_ms];
                e1[u] = (OUT_PLANES && p.Cp) ? p.c_scale[T.z1 * p.c_scale_zs + (int64_t)row * p.c_scale_ms] : 1.0f;
            } else if (i < BM + BN) {
                const int col = T.n0 + i - BM;
                e0[u] = col < p.N ? p.b_scale[T.z2 * p.sBias2 + col] : 1.0f;
                e1[u] = (p.bias && col < p.N) ? p.bias[T.z2 * p.sBias2 + col] : 0.0f;
            }
        }
    };
    Tile cur;
    int64_t t_cur = next_valid(blockIdx.x, cur);
    if (t_cur < 0) return;                                   // (workgroup-uniform)
    load_epi(cur);
    set_tile(cur);
    H3_DMA(0, 0);
    if (nk > 1) H3_DMA(1, 1);

    while (true) {
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // Main loop: ONE instruction stream, two barriers per k-tile, the two waves of every SIMD half a k-tile apart
    // (waves NW/2 .. NW-1 pass one extra barrier in front of the loop and the others one behind it: barriers match by
    // count).  A wave issues its DMA instructions at the head of its first half, behind the B-fragment reads of the new
    // k-tile; group A (slot 2 kt) fetches k-tile kt + 1 into the stage of k-tile kt - 2 and waits for it in front of its
    // next top barrier; group B (slot 2 kt + 1) fetches k-tile kt + 2 into the stage of k-tile kt - 1 and waits for it
    // in front of the middle barrier of its k-tile kt + 1 (counted: its next DMA instructions stay in flight).
    // Everything this wave has in flight - its share of k-tiles 0 and 1 and, from the second tile on, the previous
    // epilogue's stores (loads and stores share the counter and do not retire in order with each other) - has landed:
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (grpB) __builtin_amdgcn_s_barrier();
    int st = 0;                                              // stage of k-tile kt
    for (int kt = 0; kt < nk; ++kt) {
        if (!grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // A: its share of k-tile kt (issued one k-tile ago)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int rkt = kt + 1 + grpB;
        const bool refill = rkt < nk && (grpB || kt >= 1);
        const int rst = (st + 1 + grpB) % NST;
        const unsigned short* img = smem3 + st * STAGE;
        f16x8 bf[TN][NPL];
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int row = wn0 + nt * 32 + l31;
            const int off = row * BK + ((h ^ ((row >> 3) & 1)) << 3);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) bf[nt][pl] = *reinterpret_cast<const f16x8*>(&img[NPL * A_PLANE + pl * B_PLANE + off]);
        }
        if (refill) {
#pragma unroll
            for (int d = 0; d < NDMA; ++d) {
                const int i = d / NPL, pl = d % NPL;
                if (live[i]) {
                    const bool isA_ = (wave + NW * i) < CFG::A_INSTR;
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_global_load_lds(H3_ADDR(sbase[i] + (pl * pstride[i] + koff(i, rkt)), voff[i]),
                        (lds_ptr3)(smem3 + rst * STAGE + ldsoff[i] + pl * (isA_ ? A_PLANE : B_PLANE)), 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        f16x8 afp[NPL];                                      // A fragments of m-tile HM, fetched in the first half
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            f16x8 af[NPL];
            if (mt == HM) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) af[pl] = afp[pl];
            } else {
                const int row = wm0 + mt * 32 + l31;
                const int off = row * BK + ((h ^ ((row >> 3) & 1)) << 3);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) af[pl] = *reinterpret_cast<const f16x8*>(&img[pl * A_PLANE + off]);
            }
            if (mt == HM - 1) {
                const int row = wm0 + HM * 32 + l31;
                const int off = row * BK + ((h ^ ((row >> 3) & 1)) << 3);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) afp[pl] = *reinterpret_cast<const f16x8*>(&img[pl * A_PLANE + off]);
            }
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                f32x16 c = acc[mt][nt];                      // the two cross terms (2^-11 of the leading one) first
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[nt][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[nt][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[nt][0], c, 0, 0, 0);
                acc[mt][nt] = c;
            }
            if (mt == HM - 1) {
                if (grpB) {                                  // (the count is this wave's own: the last instruction slot may be empty)
                    if (!refill) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (live[IPW - 1]) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL * IPW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL * (IPW - 1) > 0 ? NPL * (IPW - 1) : 0) : "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        st = st == NST - 1 ? 0 : st + 1;
    }
    if (!grpB) __builtin_amdgcn_s_barrier();

    // ---- epilogue ----
    // MFMA layout: lane l31 = column, register e -> row (e & 3) + 8 (e >> 2) + 4 h of a 32 x 32 tile.  Each tile is
    // transposed through a wave-private LDS patch so that one store instruction writes FULL LINES: lane -> 4 consecutive
    // columns 4 (lane & 7), rows 8 q + (lane >> 3) for q = 0..3, i.e. instruction q covers 8 rows x 128 bytes.  (First
    // version: lane -> one row, 16 columns, four 16-byte stores: every instruction scattered 64 pieces of 16 bytes over 32
    // lines, and the L1 -> L2 write path took them one piece at a time: 10-12 us of stores per 256 x 256 tile, a sixth of a
    // K = 768 tile, measured by knocking the stores out.)
    // The counter of a wave's vector-memory operations retires IN ORDER, loads and stores alike: the residual of the next
    // sub-tile is fetched in front of the current one's stores, and the next tile's first DMA instructions go out in
    // front of the first store.
    const Tile done = cur;
    // fp32 output: 4 columns per lane (8 lanes = one 128-byte line of a row, 8 rows per instruction).  Planes only: 8 columns
    // per lane (16 bytes per plane: 2 lanes = one 32-byte panel row, 16 rows per instruction = 512 contiguous bytes per panel).
    constexpr int CW = (OUT_PLANES && !OUT_F32) ? 8 : 4, LPR = 32 / CW, RPI = 64 / LPR, NQ = 32 / RPI;
    const int lr = lane / LPR, c4 = CW * (lane % LPR);
    float4 r4[4];                                        // residual of the sub-tile in hand (fp32 output only: CW = 4)
    auto load_r = [&](int mt, int nt) {
        const int tn = done.n0 + wn0 + nt * 32, tm = done.m0 + wm0 + mt * 32;
        const float* Rt = p.R + done.z * p.sR + (int64_t)tm * p.ldr + tn + c4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = (tn + c4) < p.N && (tm + 8 * q + lr) < done.Mz;
            r4[q] = (ok && p.R) ? *reinterpret_cast<const float4*>(Rt + (8 * q + lr) * (int)p.ldr) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
        const int i = tid + u * CFG::THREADS;
        if (i < BM + BN) { tab0[i] = e0[u]; tab1[i] = e1[u]; }
    }
    if (HAS_R) load_r(0, 0);

    // ---- next tile: its first two k-tiles travel while this tile's epilogue runs ----
    __syncthreads();                                     // every wave is done reading the stages; the tables are filed
    t_cur = next_valid(t_cur + gridDim.x, cur);
    if (t_cur >= 0) {
        set_tile(cur);
        load_epi(cur);
        H3_DMA(0, 0);
        if (nk > 1) H3_DMA(1, 1);
    }

    {
    const int m0 = done.m0, n0 = done.n0, Mz = done.Mz;
    const int64_t z = done.z, z1 = done.z1, z2 = done.z2;
    static_assert(!HAS_R || CW == 4, "the residual comes with the fp32 output");
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int tm = m0 + wm0 + mt * 32;
        float asi[NQ], csc[NQ], row_amax[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            asi[q] = p.alpha * pow2_inverse(tab0[wm0 + mt * 32 + RPI * q + lr]);
            csc[q] = tab1[wm0 + mt * 32 + RPI * q + lr];
            row_amax[q] = 0.0f;
        }
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int tn = n0 + wn0 + nt * 32;
            const int gc = tn + c4;
            const bool c_ok = gc < p.N;                  // N % 16 == 0 and c4 % CW == 0: the lane's columns are valid together
            float bsi[CW], bias4[CW];                    // 1 / s_b (a power of two) and the bias of this lane's columns
#pragma unroll
            for (int j4 = 0; j4 < CW / 4; ++j4) {
                const float4 bs4 = *reinterpret_cast<const float4*>(tab0 + BM + wn0 + nt * 32 + c4 + 4 * j4);
                const float4 bi4 = *reinterpret_cast<const float4*>(tab1 + BM + wn0 + nt * 32 + c4 + 4 * j4);
                bsi[4 * j4] = pow2_inverse(bs4.x); bsi[4 * j4 + 1] = pow2_inverse(bs4.y);
                bsi[4 * j4 + 2] = pow2_inverse(bs4.z); bsi[4 * j4 + 3] = pow2_inverse(bs4.w);
                bias4[4 * j4] = bi4.x; bias4[4 * j4 + 1] = bi4.y; bias4[4 * j4 + 2] = bi4.z; bias4[4 * j4 + 3] = bi4.w;
            }
            constexpr int LAST = TM * TN - 1;
            const int sub = mt * TN + nt;
#pragma unroll
            for (int e = 0; e < 16; ++e) patch[(4 * h + (e & 3) + 8 * (e >> 2)) * PATCH_LD + l31] = acc[mt][nt][e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float v[NQ][CW];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j4 = 0; j4 < CW / 4; ++j4) {
                    const float4 t4 = *reinterpret_cast<const float4*>(patch + (RPI * q + lr) * PATCH_LD + c4 + 4 * j4);
                    v[q][4 * j4] = t4.x; v[q][4 * j4 + 1] = t4.y; v[q][4 * j4 + 2] = t4.z; v[q][4 * j4 + 3] = t4.w;
                }
            __builtin_amdgcn_wave_barrier();             // the patch may be overwritten by the next tile
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float rv[4] = {0.f, 0.f, 0.f, 0.f};
                if (HAS_R) { rv[0] = r4[q % 4].x; rv[1] = r4[q % 4].y; rv[2] = r4[q % 4].z; rv[3] = r4[q % 4].w; }
#pragma unroll
                for (int j = 0; j < CW; ++j) {
                    float y = v[q][j] * (asi[q] * bsi[j]) + bias4[j];       // asi * bsi: a product of powers of two (and alpha)
                    if (HAS_R) y += rv[j % 4];
                    if (ACT == ACT_SILU) y = y / (1.0f + expf(-y));
                    v[q][j] = y;
                    if (ACT == ACT_GELU && (j & 1)) {                       // two elements per packed instruction (gemm_f16x3.h)
                        const gelu_f32x2 g2 = gelu_pair(gelu_f32x2{v[q][j - 1], v[q][j]});
                        v[q][j - 1] = g2.x; v[q][j] = g2.y;
                    }
                    if (ACT != ACT_NONE && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four erf / exp chains at a time: registers
                }
            }
            if (HAS_R && sub < LAST) load_r((sub + 1) / TN, (sub + 1) % TN);   // the next residual, in front of this sub-tile's stores
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int row = tm + RPI * q + lr;
                const bool ok = c_ok && row < Mz;
                if (p.amax_out && ok && gc >= p.amax_col_min) {
#pragma unroll
                    for (int j = 0; j < CW; ++j) row_amax[q] = fmaxf(row_amax[q], fabsf(v[q][j]));
                }
                if (ok) {
                    if (OUT_F32 && p.C) *reinterpret_cast<float4*>(p.C + (done.coff >= 0 ? done.coff : z1 * p.sC) + z2 * p.sC2 + (int64_t)row * p.ldc + gc) =
                        make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
                    if (OUT_PLANES && p.Cp) {
                        const float cs = csc[q];
                        unsigned short hh[CW], ll[CW];
#pragma unroll
                        for (int j = 0; j < CW; ++j) split2(v[q][j] * cs, hh[j], ll[j]);
                        // row-major [M][ldcp], or the k16 panels of the next GEMM's A (panel gc / 16, row, k = gc % 16)
                        unsigned short* Pt = p.cp_panel ? p.Cp + (int64_t)(gc >> 4) * ((int64_t)p.M * 16) + (int64_t)row * 16 + (gc & 15)
                                                        : p.Cp + z * p.sCp + (int64_t)row * p.ldcp + gc;
                        if (CW == 8) {
                            *reinterpret_cast<uint4*>(Pt) = make_uint4(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16),
                                                                      hh[4 % CW] | ((unsigned)hh[5 % CW] << 16), hh[6 % CW] | ((unsigned)hh[7 % CW] << 16));
                            *reinterpret_cast<uint4*>(Pt + p.c_plane) = make_uint4(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16),
                                                                                  ll[4 % CW] | ((unsigned)ll[5 % CW] << 16), ll[6 % CW] | ((unsigned)ll[7 % CW] << 16));
                        } else {
                            *reinterpret_cast<uint2*>(Pt) = make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
                            *reinterpret_cast<uint2*>(Pt + p.c_plane) = make_uint2(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16));
                        }
                    }
                }
            }
        }
        if (p.amax_out) {
            // the largest |x| this wave wrote, per slot: rows are ordered by slot, so the rows of a store instruction hold
            // one slot, seldom two: one masked wave reduction + one atomic per distinct slot
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int row = tm + RPI * q + lr;
                const bool r_ok = row < Mz;
                const int slot = !r_ok ? -1 : (int)(z1 * p.amax_zs) + (p.amax_row_slot ? p.amax_row_slot[row] : 0);
                unsigned long long todo = __ballot(r_ok);
                while (todo) {
                    const int first = __ffsll((long long)todo) - 1;
                    const int s0 = __shfl(slot, first, 64);
                    const bool mine = r_ok && slot == s0;
                    float mx = mine ? row_amax[q] : 0.0f;
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                    if (lane == first && mx > 0.0f) atomicMax(p.amax_out + s0, __float_as_uint(mx));
                    todo &= ~__ballot(mine);
                }
            }
        }
    }
    }
    if (t_cur < 0) break;
    }
#undef H3_DMA
}

// ---- operand preparation ------------------------------------------------------------------------------------------
// one wave per row: scale from the exact row maximum (+ optional Euclidean norm and tensor-wide maxima)
__global__ __launch_bounds__(256) void f16x2_row_scales_kernel(const float* __restrict__ src, int64_t rows, int K, int64_t ld,
                                                               float* __restrict__ scale, float* __restrict__ norm2,
                                                               unsigned* __restrict__ stat_max) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* x = src + row * ld;
    float mx = 0.0f;
    double ss = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float v = x[k];
        mx = fmaxf(mx, fabsf(v));
        ss += (double)v * (double)v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    ss = wave_sum_f64(ss);
    if (lane == 0) {
        scale[row] = f16x2_scale_for_bound(mx);
        const float nn = (float)sqrt(ss) * 1.0000002f;           // rounded up: the norm is used as a bound
        if (norm2) norm2[row] = nn;
        if (stat_max) {
            atomicMax(stat_max + 0, __float_as_uint(nn));
            atomicMax(stat_max + 1, __float_as_uint(mx));
        }
    }
}

// a thread takes 16 consecutive k of one row (64 bytes in, 32 bytes per plane out); lanes take consecutive rows, so in the
// panel layout a wave writes 2 KiB contiguous per plane
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* __restrict__ src, int64_t rows, int K, int64_t ld,
                                                          const float* __restrict__ scale, int scale_stride,
                                                          unsigned short* __restrict__ planes, int64_t plane_stride, int panels) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int kb = blockIdx.y;                                       // panel
    const float s = scale[row * scale_stride];
    const float4* s4 = reinterpret_cast<const float4*>(src + row * ld + 16 * kb);
    unsigned short h[16], l[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = s4[q];
        split2(v.x * s, h[4 * q], l[4 * q]); split2(v.y * s, h[4 * q + 1], l[4 * q + 1]);
        split2(v.z * s, h[4 * q + 2], l[4 * q + 2]); split2(v.w * s, h[4 * q + 3], l[4 * q + 3]);
    }
    auto pk = [](const unsigned short* s_, int o) {
        return make_uint4(s_[o] | ((unsigned)s_[o + 1] << 16), s_[o + 2] | ((unsigned)s_[o + 3] << 16),
                          s_[o + 4] | ((unsigned)s_[o + 5] << 16), s_[o + 6] | ((unsigned)s_[o + 7] << 16));
    };
    unsigned short* d = panels ? planes + (int64_t)kb * rows * 16 + row * 16 : planes + row * K + 16 * kb;
    *reinterpret_cast<uint4*>(d) = pk(h, 0); *reinterpret_cast<uint4*>(d + 8) = pk(h, 8);
    *reinterpret_cast<uint4*>(d + plane_stride) = pk(l, 0); *reinterpret_cast<uint4*>(d + plane_stride + 8) = pk(l, 8);
}

__global__ __launch_bounds__(256) void scale_from_bound_kernel(const unsigned* __restrict__ amax_bits, int64_t n,
                                                               const float* __restrict__ factor_dev, float factor_host,
                                                               const float* __restrict__ add_dev, float* __restrict__ scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float f = factor_host * (factor_dev ? *factor_dev : 1.0f);
    const float a = add_dev ? *add_dev : 0.0f;
    scale[i] = f16x2_scale_for_bound(__uint_as_float(amax_bits[i]) * f + a);
}

int launch_f16x2_row_scales(const float* src, int64_t rows, int K, int64_t ld, float* scale, float* norm2, unsigned* stat_max,
                            hipStream_t stream) {
    RSAF_CHECK_ARG(rows >= 0 && K >= 0 && ld >= K, "bad shape");
    if (rows == 0) return RSAF_OK;
    RSAF_CHECK_ARG(src && scale, "NULL pointer");
    RSAF_CHECK_ARG((rows + 3) / 4 <= 0x7fffffffLL, "too many rows");
    hipLaunchKernelGGL(f16x2_row_scales_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, src, rows, K, ld, scale, norm2,
                       stat_max);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int launch_split_f16x2(const float* src, int64_t rows, int K, int64_t ld, const float* scale, int scale_stride, uint16_t* planes,
                       int64_t plane_stride, int panels, hipStream_t stream) {
    RSAF_CHECK_ARG(rows >= 0 && K >= 0 && K % 16 == 0 && ld % 4 == 0 && ld >= K, "K must be a non-negative multiple of 16, ld a multiple of 4");
    if (rows == 0 || K == 0) return RSAF_OK;
    RSAF_CHECK_ARG(src && planes && scale, "NULL pointer");
    RSAF_CHECK_ARG(plane_stride % 8 == 0 && K / 16 <= 65535 && (scale_stride == 0 || scale_stride == 1),
                   "plane stride must be a multiple of 8 elements; at most 65535 panels; scale stride 0 or 1");
    RSAF_CHECK_ARG((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(planes) & 15) == 0, "16-byte alignment");
    ProfScope prof("split_f16x2", stream, 0.0, 8.0 * (double)rows * K);
    hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)((rows + 255) / 256), (unsigned)(K / 16)), dim3(256), 0, stream, src, rows, K,
                       ld, scale, scale_stride, reinterpret_cast<unsigned short*>(planes), plane_stride, panels);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int launch_scale_from_bound(const unsigned* amax_bits, int64_t n, const float* factor_dev, float factor_host, const float* add_dev,
                            float* scale, hipStream_t stream) {
    if (n <= 0) return RSAF_OK;
    RSAF_CHECK_ARG(amax_bits && scale, "NULL pointer");
    hipLaunchKernelGGL(scale_from_bound_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, amax_bits, n, factor_dev,
                       factor_host, add_dev, scale);
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

int launch_gemm_f16x3(const GemmH3Params& p, hipStream_t stream, const char* tag) {
    RSAF_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0 && p.nz >= 0, "negative dimension");
    if (p.M == 0 || p.N == 0 || p.nz == 0) return RSAF_OK;
    RSAF_CHECK_ARG(p.A && p.B && (p.C || p.Cp), "NULL operand");
    RSAF_CHECK_ARG(p.a_scale && p.b_scale && (!p.Cp || p.c_scale), "every plane operand needs its scales");
    RSAF_CHECK_ARG(p.K % 16 == 0 && p.K >= 16, "K must be a positive multiple of 16");
    RSAF_CHECK_ARG(p.N % 16 == 0, "N must be a multiple of 16 (16-byte epilogue stores)");
    RSAF_CHECK_ARG(p.ldc % 8 == 0 && p.ldcp % 8 == 0 && p.ldr % 4 == 0 && p.c_plane % 8 == 0 && p.sC % 4 == 0 && p.sCp % 8 == 0 &&
                   p.sR % 4 == 0 && p.sC2 % 4 == 0, "ldc / ldr / output strides must keep 16-byte alignment");
    RSAF_CHECK_ARG(p.lda % 8 == 0 && p.ldb % 8 == 0 && p.a_plane % 8 == 0 && p.b_plane % 8 == 0 && p.sA % 8 == 0 && p.sA2 % 8 == 0 &&
                   p.sB2 % 8 == 0, "lda, ldb, the plane strides and the batch strides must be multiples of 8 elements");
    RSAF_CHECK_ARG((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(p.b_scale) & 15) == 0 && p.sBias2 % 4 == 0, "A, B and b_scale must be 16-byte aligned");
    RSAF_CHECK_ARG(!p.R || p.ldr > 0, "residual needs ldr");
    RSAF_CHECK_ARG(p.act >= 0 && p.act <= 2, "act must be 0 (none), 1 (gelu) or 2 (silu)");
    RSAF_CHECK_ARG(p.nz <= 65535, "at most 65535 batches per launch");
    RSAF_CHECK_ARG(!p.cp_panel || p.nz == 1, "the panel layout of the plane output is for unbatched operands");
    RSAF_CHECK_ARG(!p.a_panel || p.nz == 1 || p.a_panel_rows > 0, "a batched A in panels needs a_panel_rows");
    RSAF_CHECK_ARG(p.a_tap_panels == 0 || (p.a_panel && p.a_panel_rows > 0 && p.a_tap_panels > 0 && (p.K / 16) % p.a_tap_panels == 0),
                   "a_tap_panels: A in panels with a_panel_rows, K = taps x a_tap_panels x 16");
    RSAF_CHECK_ARG(p.ldc < (1 << 24) && p.ldcp < (1 << 24) && p.ldr < (1 << 24), "leading dimensions must be below 2^24");
    RSAF_CHECK_ARG(!p.R || p.C, "the residual comes with the fp32 output");
    RSAF_CHECK_ARG(p.nz2 <= 1 || (p.C && !p.Cp && !p.R && (!p.a_panel || p.a_panel_rows > 0) && p.nz % p.nz2 == 0),
                   "two-level batches: fp32 output only, no residual, A row-major or panels with a_panel_rows, nz a multiple of nz2");
    // algorithmic FLOPs of the contraction (2 M N K); the matrix pipe executes three fp16 products per term
    ProfScope prof(tag ? tag : "gemm_f16x3", stream, 2.0 * p.M * (double)p.N * p.K * p.nz, 0.0);
    using CfgA = H3Cfg<4, 2, 2, 4, 3, 1>;                // 256 x 256
    using CfgN = H3Cfg<2, 1, 4, 2, 3, 1>;                // 256 x 64: N <= 64 (the 48-wide groups of the positional convolution)
    using CfgM = H3Cfg<4, 2, 4, 2, 3, 1>;                // 512 x 128: N <= 128 (the CNN-LSTM's 128 channels): the wave tile of the
                                                         // 256 x 256 configuration (24 MFMAs per k-tile and wave; a 256 x 128 tile's 12
                                                         // left the main loop bound by its barriers and DMA issue: 110 TFLOP/s-equivalent)
#define H3_LAUNCH_CFG(CFG, ACT, F32, PL, HR)                                                                            \
    do {                                                                                                                \
        RSAF_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_f16x3_kernel<CFG, ACT, F32, PL, HR>,                        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES));                 \
        const int64_t tiles = (int64_t)((p.N + CFG::BN - 1) / CFG::BN) * ((p.M + CFG::BM - 1) / CFG::BM) * p.nz;        \
        RSAF_CHECK_ARG(tiles <= 0x7fffffffLL, "too many tiles");                                                        \
        hipLaunchKernelGGL((gemm_f16x3_kernel<CFG, ACT, F32, PL, HR>), dim3((unsigned)std::min<int64_t>(tiles, persistent_wgs)), \
                           dim3(CFG::THREADS), CFG::LDS_BYTES, stream, pp);                                              \
    } while (0)
#define H3_LAUNCH(ACT, F32, PL, HR)                                                                                     \
    do {                                                                                                                \
        if (p.N <= 64) H3_LAUNCH_CFG(CfgN, ACT, F32, PL, HR);                                                            \
        else H3_LAUNCH_CFG(CfgA, ACT, F32, PL, HR);                                                                      \
    } while (0)
    // the CNN-LSTM's shapes (N = 128 channels: the 256 x 128 tile) take three tile configurations
#define H3_LAUNCH3(ACT, F32, PL, HR)                                                                                    \
    do {                                                                                                                \
        if (p.N <= 64) H3_LAUNCH_CFG(CfgN, ACT, F32, PL, HR);                                                            \
        else if (p.N <= 128) H3_LAUNCH_CFG(CfgM, ACT, F32, PL, HR);                                                      \
        else H3_LAUNCH_CFG(CfgA, ACT, F32, PL, HR);                                                                      \
    } while (0)
    GemmH3Params pp = p;
    if (pp.group_m <= 0) pp.group_m = 2;
    if (pp.a_tap_panels > 0) {                               // kt / P as (kt * inv) >> 16, checked for every k-tile of this launch
        pp.a_tap_inv = 65536 / pp.a_tap_panels + 1;
        for (int kt = 0; kt < pp.K / 16; ++kt)
            RSAF_CHECK_ARG((int)(((unsigned)kt * (unsigned)pp.a_tap_inv) >> 16) == kt / pp.a_tap_panels, "a_tap_panels: K too long for the reciprocal");
    }
    // persistent workgroups, one per CU (RSAF_GEMM_WGS overrides the count: a measurement knob)
    int persistent_wgs = 256;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            persistent_wgs = cus;
        static const int forced = [] { const char* e = getenv("RSAF_GEMM_WGS"); return e ? atoi(e) : 0; }();
        if (forced > 0) persistent_wgs = forced;
    }
    const bool f32o = p.C != nullptr, plo = p.Cp != nullptr, hr = p.R != nullptr;
    // the combinations the Wav2Vec2 / CNN stages use (anything else is an argument error, not a silent fallback)
    if (p.act == ACT_NONE && f32o && !plo && !hr) H3_LAUNCH3(ACT_NONE, true, false, false);
    else if (p.act == ACT_NONE && f32o && !plo && hr) H3_LAUNCH(ACT_NONE, true, false, true);
    else if (p.act == ACT_GELU && !f32o && plo && !hr) H3_LAUNCH3(ACT_GELU, false, true, false);
    else if (p.act == ACT_SILU && !f32o && plo && !hr) H3_LAUNCH3(ACT_SILU, false, true, false);
    else if (p.act == ACT_GELU && f32o && !plo && !hr) H3_LAUNCH(ACT_GELU, true, false, false);
    else if (p.act == ACT_SILU && f32o && !plo && !hr) H3_LAUNCH(ACT_SILU, true, false, false);
    else if (p.act == ACT_GELU && f32o && !plo && hr) H3_LAUNCH3(ACT_GELU, true, false, true);
    else if (p.act == ACT_SILU && f32o && !plo && hr) H3_LAUNCH3(ACT_SILU, true, false, true);
    else if (p.act == ACT_NONE && !f32o && plo && !hr) H3_LAUNCH(ACT_NONE, false, true, false);
    else if (p.act == ACT_NONE && f32o && plo && !hr) H3_LAUNCH(ACT_NONE, true, true, false);
    else {
        set_error("launch_gemm_f16x3: unsupported combination of activation / outputs / residual");
        return RSAF_ERR_ARG;
    }
#undef H3_LAUNCH_CFG
#undef H3_LAUNCH
#undef H3_LAUNCH3
    RSAF_CHECK_HIP(hipGetLastError());
    return RSAF_OK;
}

}  // namespace rsaf

using namespace rsaf;

extern "C" int rsaf_f16x2_row_scales(const float* src, int64_t rows, int K, int64_t ld, float* scale, float* norm2,
                                     rsaf_stream_t stream) {
    return launch_f16x2_row_scales(src, rows, K, ld, scale, norm2, nullptr, (hipStream_t)stream);
}

extern "C" int rsaf_split_f16x2(const float* src, int64_t rows, int K, int64_t ld, const float* scale, int scale_stride,
                                uint16_t* planes, int64_t plane_stride, int panels, rsaf_stream_t stream) {
    return launch_split_f16x2(src, rows, K, ld, scale, scale_stride, planes, plane_stride, panels, (hipStream_t)stream);
}

extern "C" int rsaf_gemm_f16x3(const uint16_t* A_planes, int64_t a_plane_stride, const float* a_scale, int a_scale_stride,
                               const uint16_t* B_planes, int64_t b_plane_stride, const float* b_scale, float* C,
                               uint16_t* C_planes, int64_t c_plane_stride, const float* c_scale, int c_scale_stride,
                               uint32_t* amax_out, const float* bias, const float* R, int M, int N, int K, int64_t lda, int64_t ldb,
                               int64_t ldc, int64_t ldr, int act, float alpha, int a_panels, int b_panels, int c_panels,
                               rsaf_stream_t stream) {
    GemmH3Params p{};
    p.A = A_planes; p.a_plane = a_plane_stride; p.lda = a_panels ? 16 : lda; p.sA = 0;
    p.a_scale = a_scale; p.a_scale_zs = 0; p.a_scale_ms = a_scale_stride;
    p.B = B_planes; p.b_plane = b_plane_stride; p.ldb = b_panels ? 16 : ldb; p.b_scale = b_scale;
    p.C = C; p.ldc = ldc; p.sC = 0;
    p.Cp = C_planes; p.c_plane = c_plane_stride; p.ldcp = c_panels ? 16 : ldc; p.sCp = 0;
    p.c_scale = c_scale; p.c_scale_zs = 0; p.c_scale_ms = c_scale_stride;
    p.amax_out = amax_out; p.amax_zs = 0; p.amax_row_slot = nullptr; p.amax_col_min = 0;
    p.bias = bias; p.R = R; p.ldr = ldr; p.sR = 0;
    p.M = M; p.N = N; p.K = K; p.nz = 1; p.act = act; p.alpha = alpha;
    p.a_panel = a_panels != 0; p.b_panel = b_panels != 0; p.cp_panel = c_panels != 0;
    return launch_gemm_f16x3(p, (hipStream_t)stream, "gemm_f16x3");
}
