// Probe of v_mfma_f32_32x32x16_bf16 on gfx950: operand layout and issue rate (tool, not product code).
// Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_bf16_layout.hip -o tools/mfma_bf16_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(const float* A, const float* B, float* D, long long* cyc) {
    const int l = threadIdx.x;
    // hypothesis: A[i][k]: lane = i + 32*(k/8), element k%8 ; B[k][j]: lane = j + 32*(k/8), element k%8
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)A[(l % 32) * 16 + 8 * (l / 32) + e];
        b[e] = (__bf16)B[(8 * (l / 32) + e) * 32 + (l % 32)];
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int v = 0; v < 16; ++v) D[l * 16 + v] = c[v];
    f32x16 c0 = c, c1 = c, c2 = c, c3 = c;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[0] = t1 - t0;
    D[1024 + l] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    static float hA[32 * 16], hB[16 * 32], hD[1024 + 64], ref[32 * 32];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (float)((i * 3 + k * 5) % 17 - 8) * 0.25f;
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)((k * 7 + j * 2) % 13 - 6) * 0.5f;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dD; long long* dC; long long hC = 0;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD); (void)hipMalloc(&dC, 8);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dC);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); (void)hipMemcpy(&hC, dC, 8, hipMemcpyDeviceToHost);
    double e1 = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 16; ++v) {
        const int row = (v & 3) + 8 * (v >> 2) + 4 * (l / 32), col = l % 32;
        e1 = fmax(e1, fabs(hD[l * 16 + v] - ref[row * 32 + col]));
    }
    printf("layout hypothesis (A/B: lane = idx + 32*(k/8), elem k%%8; D: row=(v&3)+8(v>>2)+4(lane/32), col=lane%%32) max err %.3g\n", e1);
    printf("256 MFMA f32_32x32x16_bf16 on 4 accumulators: %.1f ticks per MFMA\n", hC / 256.0);
    return 0;
}
