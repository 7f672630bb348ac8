// Probe of the v_mfma_f32_4x4x1_16B_f32 operand layout and issue rate on gfx950 (tool, not product code).
// Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_f32_4x4_layout.hip -o tools/mfma_f32_4x4_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float float4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const float* A, const float* B, float* D, long long* cyc) {
    const int l = threadIdx.x;
    // hypothesis: 16 blocks; A[blk][i] in lane 4*blk + i ; B[blk][j] in lane 4*blk + j ; D[blk][i][j]: vgpr i, lane 4*blk + j
    const float a = A[l], b = B[l];
    float4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
    float4_t c0 = c, c1 = c, c2 = c, c3 = c;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[0] = t1 - t0;
    D[256 + l] = c0[0] + c1[1] + c2[2] + c3[3];
    // dependent chain on one accumulator
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; ++it) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[1] = t1 - t0;
    D[320 + l] = c0[0];
    // 16x16x4 f32 for comparison
    c1 = c;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[2] = t1 - t0;
    D[384 + l] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    float hA[64], hB[64], hD[512];
    for (int l = 0; l < 64; ++l) { hA[l] = 1.0f + l; hB[l] = 0.5f + 0.25f * l; }
    float *dA, *dB, *dD; long long* dC; long long hC[3] = {0, 0, 0};
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dC, 24);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dC);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hC, dC, 24, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const int blk = l / 4, j = l % 4;
        e1 = fmax(e1, fabs(hD[l * 4 + v] - hA[4 * blk + v] * hB[4 * blk + j]));     // H1: D[blk][i=v][j=lane%4]
        e2 = fmax(e2, fabs(hD[l * 4 + v] - hA[4 * blk + j] * hB[4 * blk + v]));     // H2: transposed
    }
    printf("H1 (D vgpr=i, lane=4*blk+j; A lane=4*blk+i; B lane=4*blk+j) max err %.3g\nH2 (transposed) max err %.3g\n", e1, e2);
    printf("lane 5: D = %g %g %g %g (A[4..7] = %g %g %g %g, B[5] = %g)\n", hD[20], hD[21], hD[22], hD[23], hA[4], hA[5], hA[6], hA[7], hB[5]);
    printf("256 MFMA f32 4x4x1 on 4 accumulators: %.1f ticks per MFMA; dependent chain: %.1f; f32 16x16x4 on 4 accumulators: %.1f\n",
           hC[0] / 256.0, hC[1] / 256.0, hC[2] / 256.0);
    return 0;
}
