// Sustained rate of v_mfma_f32_32x32x16_bf16 on the whole chip with no memory traffic, and the shader clock while it runs
// (clock64 = shader cycles, wall_clock64 = 100 MHz): the realistic ceiling for gemm_bf16x6 on this part.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, long long* clk, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(e + 1); }
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
    const int blocks = 256, threads = 512, iters = 4000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flop = 2.0 * 32 * 32 * 16 * 6 * 8 * (double)iters * (threads / 64) * blocks;
        printf("rep %d: %.3f ms  %.1f TFLOP/s bf16 dense   shader clock %.0f MHz (%lld cycles / %lld ticks of 100 MHz)\n", rep, ms,
               flop / ms / 1e9, (double)h[0] / (double)h[1] * 100.0, h[0], h[1]);
    }
    return 0;
}
