// Ceiling reference: back-to-back v_mfma_f32_32x32x2_f32 with (a) nothing else, (b) the GEMM's LDS
// fragment-read pattern beside it.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* __restrict__ gsrc, long gstride) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 128 * 36];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 128 * 36; i += 256) lds[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a0 = lane * 0.01f, b0 = w * 0.02f + 1.0f;
    float4 st[8];
    for (int i = 0; i < 8; ++i) st[i] = make_float4(a0, b0, a0, b0);
    const float* gp = gsrc + (long)blockIdx.x * gstride + (threadIdx.x >> 3) * 768 + (threadIdx.x & 7) * 4;
    for (int it = 0; it < iters; ++it) {
        float af[2][4], bf[2][4];
        const int boff = (it & 1) * 2 * 128 * 36;
        if (MODE >= 2) {   // stage the 'next tile': 8 ds_write_b128 per thread into the other image
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<float4*>(&lds[(boff ^ (2 * 128 * 36)) + ((threadIdx.x >> 3) + 32 * i) * 36 + (threadIdx.x & 7) * 4]) = st[i];
        }
        if (MODE >= 3) {   // refill the registers from global memory (k-tile stream of a 768-wide row panel)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                st[i] = *reinterpret_cast<const float4*>(gp + (long)(i & 3) * 32 * 768 + ((it * 32 + (i >> 2) * 0) % 768));
        }
        if (MODE >= 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 v = *reinterpret_cast<const float4*>(&lds[boff + ((w >> 1) * 64 + t * 32 + (lane & 31)) * 36 + 8 * g + 4 * (lane >> 5)]);
                    const float4 u = *reinterpret_cast<const float4*>(&lds[boff + 128 * 36 + ((w & 1) * 64 + t * 32 + (lane & 31)) * 36 + 8 * g + 4 * (lane >> 5)]);
                    af[t][0] = v.x; af[t][1] = v.y; af[t][2] = v.z; af[t][3] = v.w;
                    bf[t][0] = u.x; bf[t][1] = u.y; bf[t][2] = u.z; bf[t][3] = u.w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q >> 1][j], bf[q & 1][j], acc[q], 0, 0, 0);
            }
            if (MODE >= 2) __syncthreads();
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[q], 0, 0, 0);
        }
    }
    float s = st[0].x + st[7].w;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    static float* gsrc = nullptr;
    const long gstride = 128L * 768;
    if (!gsrc) { hipMalloc(&gsrc, 1024L * gstride * 4 + (1 << 20)); hipMemset(gsrc, 0, 1024L * gstride * 4 + (1 << 20)); }
    const int iters = MODE >= 1 ? 1000 : 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, gsrc, gstride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 64 * 4096.0;  // 64 MFMAs per iteration per wave
        printf("%s blocks=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    }
    hipFree(out);
}
int main() {
    run<0>("bare mfma, 1 wave/SIMD ", 256);
    run<0>("bare mfma, 2 waves/SIMD", 512);
    run<1>("mfma + lds frag reads, 1 wave/SIMD ", 256);
    run<1>("mfma + lds frag reads, 2 waves/SIMD", 512);
    run<2>("+ 8 ds_write_b128 + barrier / 64 mfma, 2 waves/SIMD", 512);
    run<2>("+ 8 ds_write_b128 + barrier / 64 mfma, 3 waves/SIMD", 768);
    run<3>("+ 8 global_load_dwordx4 / 64 mfma, 2 waves/SIMD", 512);
    run<3>("+ 8 global_load_dwordx4 / 64 mfma, 3 waves/SIMD", 768);
    return 0;
}
