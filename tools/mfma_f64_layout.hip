// Probe of the v_mfma_f64_16x16x4_f64 operand layout on gfx950 (tool, not product code).
// Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_f64_layout.hip -o tools/mfma_f64_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D, long long* cyc) {
    const int l = threadIdx.x;
    // hypothesis: A[i][k] in lane i + 16k ; B[k][j] in lane j + 16k
    const double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
    // throughput: 256 back-to-back independent MFMAs on 4 accumulators
    double4_t c0 = c, c1 = c, c2 = c, c3 = c;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[0] = t1 - t0;
    D[256 + l] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    double hA[64], hB[64], hD[512], ref[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1.0 + i + 0.01 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 2.0 + 0.5 * j - 0.25 * k * k;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD; long long* dC; long long hC = 0;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dC, 8);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dC);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(&hC, dC, 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        e1 = fmax(e1, fabs(hD[l * 4 + v] - ref[(4 * (l / 16) + v) * 16 + l % 16]));   // H1: row = 4*(l/16)+v
        e2 = fmax(e2, fabs(hD[l * 4 + v] - ref[((l / 16) + 4 * v) * 16 + l % 16]));   // H2: row = l/16 + 4v
    }
    printf("H1 (row=4*(lane/16)+v, col=lane%%16) max err %.3g\nH2 (row=lane/16+4v, col=lane%%16) max err %.3g\n", e1, e2);
    printf("256 MFMA f64 16x16x4: %lld cycles (readcyclecounter ticks) -> %.1f per MFMA\n", hC, hC / 256.0);
    return 0;
}
