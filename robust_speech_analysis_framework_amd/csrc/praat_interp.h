// Praat's NUM_interpolate_sinc (published source, melder/NUMinterpol.cpp) for the gfx950 resamplers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rsaf {

// Value of the samples y[0 .. n) at Praat's 1-based real index x: the depth is cut to the samples that exist on either side
// (0 -> nearest sample, 1 -> linear, 2 -> cubic), otherwise a sinc at the rate of y under a raised cosine that reaches zero
// one sample beyond the outermost sample used on each side.  The window angle advances by a fixed step per sample and is
// rotated instead of evaluated.
template <typename SRC>
__device__ inline double praat_interpolate_sinc(const SRC* __restrict__ y, int64_t n, double x, int depth) {
    constexpr double PI_ = 3.14159265358979323846;
    const int64_t midleft = (int64_t)floor(x), midright = midleft + 1;
    if (x > (double)n) return (double)y[n - 1];
    if (x < 1.0) return (double)y[0];
    if (x == (double)midleft) return (double)y[midleft - 1];
    int64_t md = depth;
    if (md > midright - 1) md = midright - 1;
    if (md > n - midleft) md = n - midleft;
    if (md <= 0) return (double)y[(int64_t)floor(x + 0.5) - 1];
    if (md == 1) return (double)y[midleft - 1] + (x - (double)midleft) * ((double)y[midright - 1] - (double)y[midleft - 1]);
    if (md == 2) {
        const double yl = (double)y[midleft - 1], yr = (double)y[midright - 1];
        const double dyl = 0.5 * (yr - (double)y[midleft - 2]), dyr = 0.5 * ((double)y[midright] - yl);
        const double fil = x - (double)midleft, fir = (double)midright - x;
        return yl * fir + yr * fil - fil * fir * (0.5 * (dyr - dyl) + (fil - 0.5) * (dyl + dyr - 2.0 * (yr - yl)));
    }
    const int64_t left = midright - md, right = midleft + md;
    double res = 0.0;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const double a0 = PI_ * (side == 0 ? x - (double)midleft : (double)midright - x);
        const double span = side == 0 ? x - (double)left + 1.0 : (double)right - x + 1.0;
        double halfsina = 0.5 * sin(a0), a = a0;
        double ws, wc, ds, dc;
        sincos(a0 / span, &ws, &wc);                 // the window angle advances by pi / span per sample
        sincos(PI_ / span, &ds, &dc);
        const SRC* p = y + (side == 0 ? midleft - 1 : midright - 1);
        const int64_t step = side == 0 ? -1 : 1;
        for (int64_t k = 0; k < md; ++k) {
            res += (double)p[k * step] * (halfsina / a * (1.0 + wc));
            a += PI_;
            halfsina = -halfsina;
            const double c2 = wc * dc - ws * ds, s2 = ws * dc + wc * ds;
            wc = c2; ws = s2;
        }
    }
    return res;
}

}  // namespace rsaf
