#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
// all data = 1.0 (e4m3 0x38).  exp_mode 0: A-scale byte b of lane l = 127 + (l>>4)*4 + b; 1: = 127 + (l&15)
template <int OPSEL, int WHICH>
__global__ void k(float* out, int exp_mode) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = 0x38383838; bv[j] = 0x38383838; }
    unsigned s = 0;
    for (int b = 0; b < 4; ++b) { unsigned e = exp_mode == 0 ? 127 + (lane >> 4) * 4 + b : 127 + (lane & 15); s |= e << (8 * b); }
    const int one = 0x7f7f7f7f;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (WHICH == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, OPSEL, (int)s, 0, one);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, one, OPSEL, (int)s);
    for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}
template <int OPSEL, int WHICH> void run(float* d, int mode) {
    k<OPSEL, WHICH><<<1, 64>>>(d, mode);
    float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    printf("%s scale varies, opsel %d, mode %d: D[m][n]/32 as hex bitsets (bit e set = a block scaled by 2^e) -- rows m=0,1,5,15 cols n=0,1,5,15:\n", WHICH ? "B" : "A", OPSEL, mode);
    for (int m : {0, 1, 5, 15}) { for (int n : {0, 1, 5, 15}) printf("  %8lx", (unsigned long)(h[m * 16 + n] / 32.0)); printf("\n"); }
}
int main() {
    float* d; hipMalloc(&d, 1024);
    run<0, 0>(d, 0); run<1, 0>(d, 0); run<2, 0>(d, 0); run<3, 0>(d, 0);
    run<0, 0>(d, 1); run<1, 0>(d, 1);
    run<0, 1>(d, 0); run<1, 1>(d, 0); run<0, 1>(d, 1);
    return 0;
}
