// Host replay of csrc/wave_fft.h: the lane functions run for the 64 lanes of an emulated wavefront, phase by phase, against
// an LDS image that records every access.  Checks (1) the transform against a direct DFT in long double, (2) the whole
// autocorrelation chain (transform, paired spectrum step, transform) against the direct circular autocorrelation and the
// cross-correlation chain against the direct sum, (3) that no ds_write_b64 has a bank conflict within its 16-lane groups and
// no ds_read_b64 within its 32-lane groups (MI355X: writes bank on (addr / 4) mod 32, reads on (addr / 4) mod 64).
// Prints one "ok ..." line per size and check; exit status 0 when everything holds.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wave_fft.h"

using namespace rsaf::wfft;

struct Access { bool store; int idx; };
struct TraceMem {
    std::vector<double>* image;
    std::vector<Access>* log;
    void chk(int i) { if (i < 0 || i >= (int)image->size()) { std::printf("LDS index out of range %d\n", i); std::exit(2); } }
    void st(int i, double v) { chk(i); (*image)[i] = v; log->push_back({true, i}); }
    void st_if(bool on, int i, double v) { if (on) st(i, v); else log->push_back({true, -1}); }   // lane masked off
    double ld(int i) { chk(i); log->push_back({false, i}); return (*image)[i]; }
};
static const long double PI = 3.14159265358979323846264338327950288L;

static int worst_conflict(const std::vector<std::vector<Access>>& logs) {
    int worst = 1;
    const size_t n = logs[0].size();
    for (int l = 1; l < 64; ++l) if (logs[l].size() != n) { std::printf("lanes disagree on the access count\n"); std::exit(2); }
    for (size_t a = 0; a < n; ++a) {
        const bool st = logs[0][a].store;
        const int group = st ? 16 : 32, slots = st ? 16 : 32;           // 8-byte slots per bank row
        for (int g0 = 0; g0 < 64; g0 += group) {
            std::vector<std::vector<int>> seen(slots);
            for (int l = g0; l < g0 + group; ++l) {
                if (logs[l][a].store != st) { std::printf("lanes disagree on the access kind\n"); std::exit(2); }
                if (logs[l][a].idx < 0) continue;
                auto& s = seen[logs[l][a].idx % slots];
                if (std::find(s.begin(), s.end(), logs[l][a].idx) == s.end()) s.push_back(logs[l][a].idx);   // same address: broadcast
            }
            for (auto& s : seen) worst = std::max(worst, (int)s.size());
        }
    }
    return worst;
}

struct Wave {
    std::vector<double> image;
    std::vector<std::vector<Access>> logs;
    explicit Wave(int doubles) : image(doubles, 0.0), logs(64) {}
    TraceMem mem(int l) { return TraceMem{&image, &logs[l]}; }
};

static cplx root(long long num, long long den) {        // exp(-2 pi i num / den)
    const long double a = -2.0L * PI * (long double)(num % den) / (long double)den;
    return cplx{(double)cosl(a), (double)sinl(a)};
}

template <int R>
static void fft_replay(cplx (*v)[R], Wave& w) {
    using P = Plan<R>;
    static cplx u[64][R];
    for (int l = 0; l < 64; ++l) { dft_bitrev<R>(v[l]); twiddle_bitrev<R>(v[l], root(l, P::S)); }
    for (int part = 0; part < 2; ++part) {
        for (int l = 0; l < 64; ++l) { auto m = w.mem(l); x1_store<R>(v[l], m, l, part); }
        for (int l = 0; l < 64; ++l) { auto m = w.mem(l); x1_load<R>(u[l], m, l, part); }
    }
    for (int l = 0; l < 64; ++l) { dft_bitrev<R>(u[l]); twiddle_bitrev<R>(u[l], root(l % P::L2, 64)); }
    for (int part = 0; part < 2; ++part) {
        for (int l = 0; l < 64; ++l) { auto m = w.mem(l); x2_store<R>(u[l], m, l, part); }
        for (int l = 0; l < 64; ++l) { auto m = w.mem(l); x2_load<R>(v[l], m, l, part); }
    }
    for (int l = 0; l < 64; ++l) { pass_c<R>(v[l], u[l]); for (int j = 0; j < R; ++j) v[l][j] = u[l][j]; }
}

static unsigned long long rng_state = 88172645463325252ULL;
static double rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) / 9007199254740992.0 - 0.5; }

template <int R>
static int check_fft() {
    using P = Plan<R>;
    constexpr int S = P::S;
    std::vector<cplx> x(S), ref(S);
    for (auto& e : x) e = cplx{rnd(), rnd()};
    for (int k = 0; k < S; ++k) {
        long double sr = 0, si = 0;
        for (int n = 0; n < S; ++n) {
            const long double a = -2.0L * PI * (long double)((long long)k * n % S) / S;
            const long double c = cosl(a), s = sinl(a);
            sr += x[n].x * c - x[n].y * s;
            si += x[n].x * s + x[n].y * c;
        }
        ref[k] = cplx{(double)sr, (double)si};
    }
    Wave w(P::LDS_DOUBLES);
    static cplx v[64][R];
    for (int l = 0; l < 64; ++l) for (int m = 0; m < R; ++m) v[l][m] = x[l + 64 * m];
    fft_replay<R>(v, w);
    double err = 0.0, scale = 0.0;
    for (int l = 0; l < 64; ++l)
        for (int m = 0; m < R; ++m) {
            const cplx r = ref[l + 64 * m];
            err = std::fmax(err, std::hypot(v[l][m].x - r.x, v[l][m].y - r.y));
            scale = std::fmax(scale, std::hypot(r.x, r.y));
        }
    const int conflict = worst_conflict(w.logs);
    const bool ok = err / scale < 1e-14 && conflict == 1;
    std::printf("%s fft R=%d rel_err=%.3e worst_conflict=%d lds_doubles=%d\n", ok ? "ok" : "FAIL", R, err / scale, conflict, P::LDS_DOUBLES);
    return ok ? 0 : 1;
}

// autocorrelation of a real frame of 2 S samples (the tail zero), as the pitch kernel chains the pieces
template <int R>
static int check_ac() {
    using P = Plan<R>;
    constexpr int S = P::S, N = 2 * S;
    const int nw = (N * 2) / 3 - 7;
    std::vector<double> seg(N, 0.0);
    for (int j = 0; j < nw; ++j) seg[j] = rnd();
    std::vector<long double> direct(N, 0.0L);
    for (int l = 0; l < N; ++l) for (int j = 0; j < N; ++j) direct[l] += (long double)seg[j] * seg[(j + l) % N];
    Wave w(P::LDS_DOUBLES);
    static cplx v[64][R];
    for (int l = 0; l < 64; ++l) for (int m = 0; m < R; ++m) v[l][m] = cplx{seg[2 * (l + 64 * m)], seg[2 * (l + 64 * m) + 1]};
    fft_replay<R>(v, w);
    cplx yh[64];
    for (int l = 0; l < 64; ++l) { auto m = w.mem(l); ac_spec_store<R>(v[l], m, l); }
    for (int l = 0; l < 64; ++l) { auto m = w.mem(l); yh[l] = ac_spec_pairs<R>(v[l], m, l, root(l, N)); }
    for (int l = 0; l < 64; ++l) { auto m = w.mem(l); ac_spec_load<R>(v[l], m, l, yh[l]); }
    fft_replay<R>(v, w);
    const double r0 = v[0][0].x;
    double err = 0.0;
    for (int l = 0; l < 64; ++l)
        for (int m = 0; m < R; ++m) {
            const int k = l + 64 * m;
            err = std::fmax(err, std::fabs(v[l][m].x / r0 - (double)(direct[2 * k] / direct[0])));
            err = std::fmax(err, std::fabs(-v[l][m].y / r0 - (double)(direct[2 * k + 1] / direct[0])));
        }
    const int conflict = worst_conflict(w.logs);
    const bool ok = err < 1e-13 && conflict == 1;
    std::printf("%s ac  R=%d max_err=%.3e worst_conflict=%d\n", ok ? "ok" : "FAIL", R, err, conflict);
    return ok ? 0 : 1;
}

// forward cross-correlation r(l) = sum_{j < nw} b[j] b[j + l] through z = a + i b, a = b on [0, nw) and zero behind
template <int R>
static int check_cc() {
    using P = Plan<R>;
    constexpr int S = P::S, M = S / 2;
    const int nw = S / 2 - 11, L = S / 4 - 3, seg_len = nw + L + 1;
    std::vector<double> b(S, 0.0);
    for (int j = 0; j < seg_len; ++j) b[j] = rnd();
    Wave w(P::LDS_DOUBLES);
    static cplx v[64][R];
    static cplx y[64][R / 2];
    for (int l = 0; l < 64; ++l) for (int m = 0; m < R; ++m) { const int j = l + 64 * m; v[l][m] = cplx{j < nw ? b[j] : 0.0, b[j]}; }
    fft_replay<R>(v, w);
    cplx yh[64];
    for (int l = 0; l < 64; ++l) { auto m = w.mem(l); cc_spec_store<R>(v[l], m, l); }
    for (int l = 0; l < 64; ++l) { auto m = w.mem(l); yh[l] = cc_spec_pairs<R>(v[l], y[l], m, l, root(l, S)); }
    for (int l = 0; l < 64; ++l) { auto m = w.mem(l); cc_spec_load<R>(y[l], m, l, yh[l]); }
    Wave w2(Plan<R / 2>::LDS_DOUBLES);
    fft_replay<R / 2>(y, w2);
    double err = 0.0;
    for (int l = 0; l < 64; ++l)
        for (int m = 0; m < R / 2; ++m) {
            const int k = l + 64 * m;
            for (int odd = 0; odd < 2; ++odd) {
                const int lag = 2 * k + odd;
                if (lag > L) continue;
                long double d = 0.0L;
                for (int j = 0; j < nw; ++j) d += (long double)b[j] * b[j + lag];
                const double got = (odd ? -y[l][m].y : y[l][m].x) / (double)S;
                err = std::fmax(err, std::fabs(got - (double)d));
            }
        }
    (void)M;
    const int conflict = std::max(worst_conflict(w.logs), worst_conflict(w2.logs));
    const bool ok = err < 1e-12 && conflict == 1;
    std::printf("%s cc  R=%d max_err=%.3e worst_conflict=%d\n", ok ? "ok" : "FAIL", R, err, conflict);
    return ok ? 0 : 1;
}

int main() {
    int bad = 0;
    bad |= check_fft<8>() | check_fft<16>() | check_fft<32>();
    bad |= check_ac<8>() | check_ac<16>() | check_ac<32>();
    bad |= check_cc<16>() | check_cc<32>();
    return bad;
}
