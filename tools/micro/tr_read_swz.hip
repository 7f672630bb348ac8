// ds_read_b64_tr_b16 on the attention kernel's swizzled V image ([128 keys][64 halfs], chunk ^= ((row >> 1) & 1) << 2):
// the transposed-read fragment against the element-by-element fragment, for every (t4, s2, ct).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(int* out, int swz) {
    __shared__ unsigned short buf[128 * 64];
    for (int i = threadIdx.x; i < 128 * 64; i += 64) {
        const int row = i / 64, col = i % 64;
        const int f = swz ? (((row >> 1) & 1) << 2) : 0;
        buf[row * 64 + 8 * ((col >> 3) ^ f) + (col & 7)] = (unsigned short)i;
    }
    __syncthreads();
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int gq = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    int bad = 0, first = -1;
    for (int t4 = 0; t4 < 4; ++t4) for (int s2 = 0; s2 < 2; ++s2) for (int ct = 0; ct < 2; ++ct) {
        unsigned short want[8], got[8];
        for (int j = 0; j < 8; ++j) {
            const int row = 32 * t4 + 16 * s2 + 8 * (j >> 2) + 4 * h + (j & 3), d = 32 * ct + l31;
            const int f = swz ? (((row >> 1) & 1) << 2) : 0;
            want[j] = buf[row * 64 + 8 * ((d >> 3) ^ f) + (d & 7)];
        }
        for (int half = 0; half < 2; ++half) {
            const int row = 32 * t4 + 16 * s2 + 8 * half + 4 * (gq >> 1) + tq;
            const int f = swz ? (((row >> 1) & 1) << 2) : 0;
            const int chunk = (4 * ct + 2 * (gq & 1) + (tp >> 1)) ^ f;
            s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(buf + row * 64 + 8 * chunk + 4 * (tp & 1)));
            for (int j = 0; j < 4; ++j) got[4 * half + j] = (unsigned short)v[j];
        }
        for (int j = 0; j < 8; ++j) if (want[j] != got[j]) { ++bad; if (first < 0) first = (t4 << 20) | (s2 << 16) | (ct << 12) | (j << 8) | 0; out[64 + 4 * lane] = want[j]; out[64 + 4 * lane + 1] = got[j]; }
    }
    out[lane] = bad;
    out[64 + 4 * lane + 2] = first;
}
int main() {
    int* d; (void)hipMalloc(&d, 1024 * 4);
    for (int swz = 0; swz < 2; ++swz) {
        (void)hipMemset(d, 0, 1024 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, swz);
        int h[1024]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int tot = 0; for (int L = 0; L < 64; ++L) tot += h[L];
        printf("swizzle %d: %d mismatching elements\n", swz, tot);
        for (int L = 0; L < 64 && tot; L += 5) printf("  lane %2d bad %3d want (r%d,c%d) got (r%d,c%d)\n", L, h[L], h[64 + 4 * L] / 64, h[64 + 4 * L] % 64, h[64 + 4 * L + 1] / 64, h[64 + 4 * L + 1] % 64);
    }
    return 0;
}
