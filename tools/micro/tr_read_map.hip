// Which element does ds_read_b64_tr_b16 hand to which lane?  LDS holds value(row, col) = 64 row + col of a [8][64] fp16
// image; lane L of 16-lane group g supplies the address of row (L & 15) >> 2, columns 16 g + 4 (L & 3) .. + 3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(int* out) {
    __shared__ unsigned short buf[8 * 64];
    for (int i = threadIdx.x; i < 8 * 64; i += 64) buf[i] = (unsigned short)i;
    __syncthreads();
    const int L = threadIdx.x, g = L >> 4, i = L & 15;
    const int off = (i >> 2) * 64 + 16 * g + 4 * (i & 3);
    s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(buf + off));
    for (int j = 0; j < 4; ++j) out[4 * L + j] = (unsigned short)v[j];
}
int main() {
    int* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int L = 0; L < 64; ++L) {
        printf("lane %2d:", L);
        for (int j = 0; j < 4; ++j) printf(" (r%d,c%2d)", h[4 * L + j] / 64, h[4 * L + j] % 64);
        printf("\n");
    }
    return 0;
}
