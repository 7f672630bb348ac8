// Complex fp64 FFT of S = 64 R points held by ONE wavefront, R points per lane (R = 8, 16, 32: S = 512, 1024, 2048),
// for the MSHDS pitch analyses (Sound: To Pitch (ac/cc) as src/mshds_extractor.py:143,178,221 of the reference call it).
//
// Layout in and out: lane l, register m holds element l + 64 m.  Three passes, all arithmetic in registers:
//   pass A   DFT_R over m (stride 64 in the sequence), twiddle W_S^(l k1)              -> y[l][k1]
//   exchange 1 through LDS: lane (k1, b) gathers y[L2 a + b][k1], a < R                   (L2 = 64 / R)
//   pass B   DFT_R over a, twiddle W_64^(b q1)                                          -> z[k1][b][q1]
//   exchange 2 through LDS: lane k1 + R qh gathers z[k1][b][g L2 + qh], g < R / L2, b < L2
//   pass C   R / L2 transforms DFT_L2 over b                                            -> X[k1 + R q1 + R^2 q2]
// and k1 + R (g L2 + qh) + R^2 q2 = lane + 64 (g + (R / L2) q2): the result is in the input layout again, so the pitch
// kernels chain transform -> spectrum -> transform without another exchange, and their global loads / stores (element
// l + 64 m from lane l) are coalesced.  Against the workgroup-wide Stockham transform this replaces
// (5 radix-4 stages of 256 threads, a workgroup barrier behind each) a frame's transform makes 2 instead of 5 trips through
// LDS and never waits at a barrier: a wave's LDS operations complete in order.
//
// The exchanges move the real parts, then the imaginary parts, through one buffer of LDS_DOUBLES doubles per wave
// (8.5 KB at S = 1024).  Strides are padded so that every ds_write_b64 is conflict-free within its 16-lane groups and
// every ds_read_b64 within its 32-lane groups (MI355X_MICROARCH.md, LDS): tests/test_wave_fft_host.py replays the
// address streams on the CPU, together with the arithmetic (the lane functions compile for the host too).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define WF_HD __host__ __device__ __forceinline__
#else
#define WF_HD inline
#endif

namespace rsaf {
namespace wfft {

struct cplx {
    double x, y;
};
WF_HD cplx operator+(cplx a, cplx b) { return cplx{a.x + b.x, a.y + b.y}; }
WF_HD cplx operator-(cplx a, cplx b) { return cplx{a.x - b.x, a.y - b.y}; }
WF_HD cplx cmul(cplx a, cplx b) { return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// W_64^j = exp(-2 pi i j / 64), j < 32: (cos, -sin)
WF_HD constexpr double w64_re(int j) {
    constexpr double t[32] = {
        1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494, 0.92387953251128675613,
        0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081, 0.70710678118654752440,
        0.63439328416364549822, 0.55557023301960222474, 0.47139673682599764856, 0.38268343236508977173,
        0.29028467725446236764, 0.19509032201612826785, 0.09801714032956060199, 0.0, -0.09801714032956060199,
        -0.19509032201612826785, -0.29028467725446236764, -0.38268343236508977173, -0.47139673682599764856,
        -0.55557023301960222474, -0.63439328416364549822, -0.70710678118654752440, -0.77301045336273696081,
        -0.83146961230254523708, -0.88192126434835502971, -0.92387953251128675613, -0.95694033573220886494,
        -0.98078528040323044913, -0.99518472667219688624};
    return t[j];
}
WF_HD constexpr double w64_im(int j) { return j == 0 ? 0.0 : -w64_re(j >= 16 ? j - 16 : 16 - j); }   // -sin = -cos(x - pi/2)

// v * W_64^t, t < 32 a compile-time constant once the caller's loops are unrolled
WF_HD cplx mul_w64(cplx v, int t) {
    constexpr double h = 0.70710678118654752440;
    if (t == 0) return v;
    if (t == 16) return cplx{v.y, -v.x};
    if (t == 8) return cplx{(v.x + v.y) * h, (v.y - v.x) * h};
    if (t == 24) return cplx{(v.y - v.x) * h, -(v.x + v.y) * h};
    return cmul(v, cplx{w64_re(t), w64_im(t)});
}

WF_HD constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
WF_HD constexpr int bitrev(int j, int bits) {           // bits <= 5, loop-free so that it folds wherever j is a constant
    return (((j & 1) << 4) | ((j & 2) << 2) | (j & 4) | ((j & 8) >> 2) | ((j & 16) >> 4)) >> (5 - bits);
}

// in-place decimation-in-frequency DFT of the R values of one lane: v[j] <- X[bitrev(j)]
// (loops in canonical form: everything unrolls and the array stays in registers)
template <int R>
WF_HD void dft_bitrev(cplx* v) {
    constexpr int LOGR = ilog2(R);
#pragma unroll
    for (int lv = 0; lv < LOGR; ++lv) {
        const int span = (R >> 1) >> lv;
#pragma unroll
        for (int bi = 0; bi < R / 2; ++bi) {
            const int j = bi & (span - 1), lo = ((bi & ~(span - 1)) << 1) | j;
            const cplx a = v[lo], b = v[lo + span];
            v[lo] = a + b;
            v[lo + span] = mul_w64(a - b, j * (32 / span));
        }
    }
}

// v[slot of k] *= w^k for k = 1 .. R - 1, where slot(k) = bitrev(k): running powers advanced by w^4 (four chains), or by
// w^2 (two chains, 8 registers fewer: the 32-point passes, which are short of registers)
template <int R>
WF_HD void twiddle_bitrev(cplx* v, cplx w1) {
    constexpr int LOGR = ilog2(R);
    if (R >= 32) {
        const cplx w2 = cmul(w1, w1);
        cplx ce = w2, co = w1;                          // ce serves k = 2 i from i = 1 on
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
            if (i > 0) v[bitrev(2 * i, LOGR)] = cmul(v[bitrev(2 * i, LOGR)], ce);
            v[bitrev(2 * i + 1, LOGR)] = cmul(v[bitrev(2 * i + 1, LOGR)], co);
            if (i + 1 < R / 2) {
                if (i > 0) ce = cmul(ce, w2);
                co = cmul(co, w2);
            }
        }
        return;
    }
    const cplx w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
    cplx c0 = w4, c1 = w1, c2 = w2, c3 = w3;          // c0 serves k = 4 i from i = 1 on
#pragma unroll
    for (int i = 0; i < R / 4; ++i) {
        if (i > 0) v[bitrev(4 * i, LOGR)] = cmul(v[bitrev(4 * i, LOGR)], c0);
        v[bitrev(4 * i + 1, LOGR)] = cmul(v[bitrev(4 * i + 1, LOGR)], c1);
        v[bitrev(4 * i + 2, LOGR)] = cmul(v[bitrev(4 * i + 2, LOGR)], c2);
        v[bitrev(4 * i + 3, LOGR)] = cmul(v[bitrev(4 * i + 3, LOGR)], c3);
        if (i + 1 < R / 4) {
            if (i > 0) c0 = cmul(c0, w4);
            c1 = cmul(c1, w4);
            c2 = cmul(c2, w4);
            c3 = cmul(c3, w4);
        }
    }
}

template <int R>
struct Plan {
    static constexpr int S = 64 * R;
    static constexpr int LOGR = ilog2(R);
    static constexpr int L2 = 64 / R;                 // radix of pass C
    static constexpr int LOGL2 = ilog2(L2);
    static constexpr int G = R / L2;                  // pass C transforms per lane
    static constexpr int X1_PITCH = 64 + L2;          // doubles between the k1 rows of exchange 1
    static constexpr int X2_PITCH = S / L2 + 16 / L2; // doubles between the b planes of exchange 2
    static constexpr int X1_DOUBLES = R * X1_PITCH, X2_DOUBLES = L2 * X2_PITCH;
    static constexpr int LDS_DOUBLES = (X1_DOUBLES > X2_DOUBLES ? X1_DOUBLES : X2_DOUBLES) > S
                                           ? (X1_DOUBLES > X2_DOUBLES ? X1_DOUBLES : X2_DOUBLES) : S;
    static_assert(R == 8 || R == 16 || R == 32, "64 R = 512, 1024 or 2048 points");
};

// The lane functions take the LDS buffer through an accessor (st / st_if / ld of one double at a double index) so that the host
// replay can record the address streams; `part` = 0: real parts, 1: imaginary parts.
template <int R, class Mem>
WF_HD void x1_store(const cplx* v, Mem& m, int lane, int part) {
    using P = Plan<R>;
#pragma unroll
    for (int j = 0; j < R; ++j) m.st(bitrev(j, P::LOGR) * P::X1_PITCH + lane, part ? v[j].y : v[j].x);
}
template <int R, class Mem>
WF_HD void x1_load(cplx* u, Mem& m, int lane, int part) {
    using P = Plan<R>;
    const int b = lane % P::L2, k1 = lane / P::L2;
#pragma unroll
    for (int a = 0; a < R; ++a) {
        const double t = m.ld(k1 * P::X1_PITCH + b + P::L2 * a);
        if (part) u[a].y = t; else u[a].x = t;
    }
}
template <int R, class Mem>
WF_HD void x2_store(const cplx* u, Mem& m, int lane, int part) {
    using P = Plan<R>;
    const int b = lane % P::L2, k1 = lane / P::L2;
#pragma unroll
    for (int j = 0; j < R; ++j) m.st(b * P::X2_PITCH + bitrev(j, P::LOGR) * R + k1, part ? u[j].y : u[j].x);
}
template <int R, class Mem>
WF_HD void x2_load(cplx* w, Mem& m, int lane, int part) {        // w[g * L2 + b]
    using P = Plan<R>;
#pragma unroll
    for (int g = 0; g < P::G; ++g) {
#pragma unroll
        for (int b = 0; b < P::L2; ++b) {
            const double t = m.ld(b * P::X2_PITCH + 64 * g + lane);
            if (part) w[g * P::L2 + b].y = t; else w[g * P::L2 + b].x = t;
        }
    }
}
// pass C: w[g * L2 + b] -> v[g + G q2]
template <int R>
WF_HD void pass_c(cplx* w, cplx* v) {
    using P = Plan<R>;
#pragma unroll
    for (int g = 0; g < P::G; ++g) {
        dft_bitrev<P::L2>(w + g * P::L2);
#pragma unroll
        for (int j = 0; j < P::L2; ++j) v[g + P::G * bitrev(j, P::LOGL2)] = w[g * P::L2 + j];
    }
}

// ---- spectrum steps of the pitch kernels between their two transforms -----------------------------------------------
// Autocorrelation (real frame of 2 S samples packed as z[j] = x[2 j] + i x[2 j + 1], Z = FFT_S z): with
// E = (Z[k] + conj Z[S-k]) / 2, O = (Z[k] - conj Z[S-k]) / 2i, w = W_2S^k the spectrum of the frame is X[k] = E + w O,
// X[S-k] = conj(E - w O), and the packed input of the transform back (stored conjugated, so that a forward transform
// inverts) is Y[k] = (s + w.y d, -w.x d), Y[S-k] = (s - w.y d, -w.x d) with s, d = |X[k]|^2 +- |X[S-k]|^2.
// The pair (k, S-k) is evaluated ONCE, by the lane that holds k < S / 2: the upper half of Z goes to LDS (two images of
// S / 2 doubles), the lane reads Z[S-k] from the slot of its partner, writes Y[S-k] back into that very slot (no other
// lane touches it) and the owners read their upper halves back.  k = 0 and k = S / 2 are their own partners (lane 0).
struct SpecAc {
    cplx yk, yp;
};
WF_HD SpecAc spec_ac(cplx zk, cplx zm, cplx w) {
    const cplx E{0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y)};
    const cplx D{0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y)};
    const cplx T = cmul(w, cplx{D.y, -D.x});
    const cplx xa = E + T, xb = E - T;
    const double pk = xa.x * xa.x + xa.y * xa.y, pm = xb.x * xb.x + xb.y * xb.y;
    const double sum = pk + pm, d = pk - pm;
    return SpecAc{cplx{sum + w.y * d, -(w.x * d)}, cplx{sum - w.y * d, -(w.x * d)}};
}
template <int R, class Mem>
WF_HD void ac_spec_store(const cplx* v, Mem& m, int lane) {
    constexpr int H = R / 2, S = 64 * R;
#pragma unroll
    for (int j = H; j < R; ++j) {
        m.st(lane + 64 * (j - H), v[j].x);
        m.st(S / 2 + lane + 64 * (j - H), v[j].y);
    }
}
// w_l = W_2S^lane.  Returns Y[S/2] (meaningful in lane 0), to be put into v[R/2] of lane 0 after ac_spec_load.
template <int R, class Mem>
WF_HD cplx ac_spec_pairs(cplx* v, Mem& m, int lane, cplx w_l) {
    constexpr int H = R / 2, S = 64 * R;
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const int k = lane + 64 * j;
        const int pp = k ? S / 2 - k : 0;
        cplx zm{m.ld(pp), m.ld(S / 2 + pp)};
        if (j == 0) zm = cplx{lane == 0 ? v[0].x : zm.x, lane == 0 ? v[0].y : zm.y};   // k = 0 pairs with itself
        const SpecAc y = spec_ac(v[j], zm, mul_w64(w_l, j * (32 / R)));
        v[j] = y.yk;
        m.st_if(k != 0, pp, y.yp.x);
        m.st_if(k != 0, S / 2 + pp, y.yp.y);
    }
    const cplx zh = v[H];                               // lane 0: Z[S/2], its own partner, w = -i
    return cplx{2.0 * (zh.x * zh.x + zh.y * zh.y), 0.0};
}
template <int R, class Mem>
WF_HD void ac_spec_load(cplx* v, Mem& m, int lane, cplx y_half) {
    constexpr int H = R / 2, S = 64 * R;
#pragma unroll
    for (int j = H; j < R; ++j) v[j] = cplx{m.ld(lane + 64 * (j - H)), m.ld(S / 2 + lane + 64 * (j - H))};
    if (lane == 0) v[H] = y_half;
}

// Cross-correlation (z = a + i b, Z = FFT_S z, M = S / 2): A[k] = (Z[k] + conj Z[S-k]) / 2, B[k] = (Z[k] - conj Z[S-k]) / 2i,
// C = conj(A) B, and the M-point packed input of the transform back (stored conjugated) is
// Y[k] = conj((C[k] + conj C[M-k]) + i conj(w) (C[k] - conj C[M-k])), w = W_S^k.  The pair (k, M-k) is evaluated once, by
// the lane that holds k < M / 2 in registers j (Z[k]) and j + R/2 (Z[M+k]); Z[S-k] and Z[M-k] come from the fourth and
// second quarter of the partner lane's registers through LDS, and Y[M-k] goes back into the slot Z[M-k] was read from.
WF_HD cplx spec_cc(cplx zk, cplx zn) {
    const cplx A{0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y)};
    const cplx B{0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x)};
    return cplx{A.x * B.x + A.y * B.y, A.x * B.y - A.y * B.x};
}
WF_HD cplx cc_y_low(cplx ck, cplx cm, cplx w) {       // Y[k]
    const cplx su{ck.x + cm.x, ck.y - cm.y}, di{ck.x - cm.x, ck.y + cm.y};
    return cplx{su.x - (w.x * di.y - w.y * di.x), -(su.y + (w.x * di.x + w.y * di.y))};
}
WF_HD cplx cc_y_high(cplx ck, cplx cm, cplx w) {      // Y[M-k]: W_S^(M-k) = -conj(w)
    const cplx su{cm.x + ck.x, cm.y - ck.y}, di{cm.x - ck.x, cm.y + ck.y};
    return cplx{su.x + (w.x * di.y + w.y * di.x), -(su.y - (w.x * di.x - w.y * di.y))};
}
template <int R, class Mem>
WF_HD void cc_spec_store(const cplx* v, Mem& m, int lane) {
    constexpr int Q = R / 4, S = 64 * R;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        m.st(lane + 64 * j, v[3 * Q + j].x);
        m.st(S / 4 + lane + 64 * j, v[3 * Q + j].y);
        m.st(S / 2 + lane + 64 * j, v[Q + j].x);
        m.st(3 * S / 4 + lane + 64 * j, v[Q + j].y);
    }
}
// w_l = W_S^lane; y[j], j < R/4, are written; returns Y[M/2] (meaningful in lane 0) for cc_spec_load
template <int R, class Mem>
WF_HD cplx cc_spec_pairs(const cplx* v, cplx* y, Mem& m, int lane, cplx w_l) {
    constexpr int Q = R / 4, H = R / 2, S = 64 * R;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int k = lane + 64 * j;
        const int pp = k ? S / 4 - k : 0;
        cplx zn{m.ld(pp), m.ld(S / 4 + pp)};
        cplx z2{m.ld(S / 2 + pp), m.ld(3 * S / 4 + pp)};
        if (j == 0) {                                     // k = 0 (lane 0) pairs with itself and with k = M
            const bool self = lane == 0;
            zn = cplx{self ? v[0].x : zn.x, self ? v[0].y : zn.y};
            z2 = cplx{self ? v[H].x : z2.x, self ? v[H].y : z2.y};
        }
        const cplx ck = spec_cc(v[j], zn), cm = spec_cc(z2, v[j + H]);
        const cplx w = mul_w64(w_l, j * (64 / R));
        y[j] = cc_y_low(ck, cm, w);
        const cplx yh = cc_y_high(ck, cm, w);
        m.st_if(k != 0, S / 2 + pp, yh.x);
        m.st_if(k != 0, 3 * S / 4 + pp, yh.y);
    }
    const cplx cs = spec_cc(v[Q], v[3 * Q]);            // lane 0: k = M/2 pairs with itself, w = -i
    return cc_y_low(cs, cs, cplx{0.0, -1.0});
}
template <int R, class Mem>
WF_HD void cc_spec_load(cplx* y, Mem& m, int lane, cplx y_half) {
    constexpr int Q = R / 4, S = 64 * R;
#pragma unroll
    for (int j = 0; j < Q; ++j) y[Q + j] = cplx{m.ld(S / 2 + lane + 64 * j), m.ld(3 * S / 4 + lane + 64 * j)};
    if (lane == 0) y[Q] = y_half;
}

#if defined(__HIPCC__)
struct LdsMem {
    double* p;
    __device__ __forceinline__ void st(int i, double v) { p[i] = v; }
    __device__ __forceinline__ void st_if(bool on, int i, double v) { if (on) p[i] = v; }
    __device__ __forceinline__ double ld(int i) const { return p[i]; }
};
// a wave's LDS operations execute in order: the fence only pins the compiler's schedule
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// forward transform of v (layout above), result in v.  w_s = W_S^lane, w_b = W_64^(lane % L2).
template <int R>
__device__ __forceinline__ void wave_fft(cplx (&v)[R], double* lds, int lane, cplx w_s, cplx w_b) {
    LdsMem m{lds};
    cplx u[R];
    dft_bitrev<R>(v);
    twiddle_bitrev<R>(v, w_s);
    x1_store<R>(v, m, lane, 0);
    wave_sync();
    x1_load<R>(u, m, lane, 0);
    wave_sync();
    x1_store<R>(v, m, lane, 1);
    wave_sync();
    x1_load<R>(u, m, lane, 1);
    wave_sync();
    dft_bitrev<R>(u);
    twiddle_bitrev<R>(u, w_b);
    x2_store<R>(u, m, lane, 0);
    wave_sync();
    x2_load<R>(v, m, lane, 0);
    wave_sync();
    x2_store<R>(u, m, lane, 1);
    wave_sync();
    x2_load<R>(v, m, lane, 1);
    wave_sync();
    pass_c<R>(v, u);
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = u[j];
}

#endif

}  // namespace wfft
}  // namespace rsaf
