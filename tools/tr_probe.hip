#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int idx;   // bf16 element index of this lane's address
    if (mode == 0) idx = (l >> 4) * 64 + (l & 15) * 4;            // group g: 128 B contiguous block, lane i at i*8 B
    else if (mode == 1) idx = (l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4;   // same thing written as row (i/4), col 4*(i%4)
    else idx = (l >> 4) * 256 + ((l & 15) >> 2) * 40 + (l & 3) * 4;                // rows with pitch 40 elements (80 B)
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + idx));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("l%02d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 4 == 3) ? "\n" : "   "); }
    }
    return 0;
}
