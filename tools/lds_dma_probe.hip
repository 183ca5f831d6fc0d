#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n * 4, 0x00020000);
    // each lane loads 16 bytes at its own offset; data lands in LDS at base + lane*16
    unsigned ofs = threadIdx.x * 16u;
    if (threadIdx.x & 1) ofs = 0x80000000u;     // out of range: expect zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, ofs, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    y[threadIdx.x * 4 + 0] = smem[threadIdx.x * 4 + 0];
    y[threadIdx.x * 4 + 1] = smem[threadIdx.x * 4 + 1];
    y[threadIdx.x * 4 + 2] = smem[threadIdx.x * 4 + 2];
    y[threadIdx.x * 4 + 3] = smem[threadIdx.x * 4 + 3];
}
int main() {
    float *x, *y; int n = 1024;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4);
    float h[1024]; for (int i = 0; i < n; ++i) h[i] = i + 1;
    hipMemcpy(x, h, n * 4, hipMemcpyHostToDevice);
    hipMemset(y, 0xff, n * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, x, y, n);
    float o[256]; hipMemcpy(o, y, 256 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 24; ++i) printf("%g ", o[i]); printf("\n");
    return 0;
}
