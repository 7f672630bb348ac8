// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950 (tool, not product code): which output lane receives the product of A's lane la
// and B's lane lb, and the instruction's issue rate.
// Build: hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_f64_4x4_probe.hip -o tools/micro/mfma_f64_4x4_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out, long long* cyc) {
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) out[la * 64 + lb] = l;
        }
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const double a = 1.0 + l, b = 0.5 * l;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[0] = t1 - t0;
    if (c0 + c1 + c2 + c3 == 12345.0) out[0] = -2;
}
int main() {
    int* d; long long* c; int h[4096]; long long hc = 0;
    hipMalloc(&d, sizeof h); hipMalloc(&c, 8);
    hipMemset(d, 0xff, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    // hypothesis: block = lane / 16; A[i][k]: lane = 16 blk + 4 k + i; B[k][j]: lane = 16 blk + 4 k + j; D[i][j]: lane = 16 blk + 4 i + j
    int bad = 0, n = 0;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
        const int o = h[la * 64 + lb];
        const int ba = la / 16, ka = (la % 16) / 4, i = la % 4, bb = lb / 16, kb = (lb % 16) / 4, j = lb % 4;
        const int want = (ba == bb && ka == kb) ? 16 * ba + 4 * i + j : -1;
        if (o != -1) ++n;
        if (o != want) { if (bad < 12) printf("la %d lb %d -> lane %d (hypothesis %d)\n", la, lb, o, want); ++bad; }
    }
    printf("pairs with a product: %d (expected 256); mismatches against the hypothesis: %d\n", n, bad);
    printf("sample: A lane 5 x B lane 6 -> lane %d; A lane 21 x B lane 22 -> lane %d; A lane 1 x B lane 2 -> %d\n", h[5 * 64 + 6], h[21 * 64 + 22], h[1 * 64 + 2]);
    printf("256 MFMA f64 4x4x4: %lld ticks -> %.2f per MFMA (readcyclecounter runs at 100 MHz: x clock / 100e6 for cycles)\n", hc, hc / 256.0);
    return 0;
}
