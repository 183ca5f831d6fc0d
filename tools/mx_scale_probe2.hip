#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, unsigned ea4, unsigned eb4, int dataA, int dataB) {
    const int lane = threadIdx.x, g = lane >> 4;
    i32x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = dataA; bv[j] = dataB; }
    const int sa = (int)(((ea4 >> (8 * g)) & 0xffu) * 0x01010101u), sb = (int)(((eb4 >> (8 * g)) & 0xffu) * 0x01010101u);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}
void run(float* d, unsigned ea4, unsigned eb4, int dA, int dB, double expect) {
    k<<<1, 64>>>(d, ea4, eb4, dA, dB);
    float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    printf("A %08x B %08x data %08x %08x: D[0][0] = %.9g D[7][9] = %.9g   expected %.9g\n", ea4, eb4, dA, dB, h[0], h[7 * 16 + 9], expect);
}
int main() {
    float* d; hipMalloc(&d, 1024);
    auto p2 = [](int e) { return __builtin_ldexp(1.0, e); };
    run(d, 0x6d776d77u, 0x7f737f73u, 0x38383838, 0x38383838, 32 * (2 * p2(-20) + 2 * p2(-18)));
    run(d, 0x77777777u, 0x73737373u, 0x38383838, 0x38383838, 128 * p2(-20));
    run(d, 0x6d6d6d6du, 0x7f7f7f7fu, 0x38383838, 0x38383838, 128 * p2(-18));
    run(d, 0x6d776d77u, 0x7f7f7f7fu, 0x38383838, 0x38383838, 64 * (p2(-8) + p2(-18)));
    run(d, 0x7f7f7f7fu, 0x7f737f73u, 0x38383838, 0x38383838, 64 * (p2(-12) + 1));
    run(d, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x38383838, 0x38383838, 128);
    // accumulation precision: data 1.0 x 1.0 in block 0 scaled 2^0, blocks 1-3 scaled 2^-e: at which e do the small blocks vanish?
    for (int e = 8; e <= 28; e += 4) run(d, 0x7f7f7f7fu ^ 0, ((127 - e) << 8 | (127 - e) << 16 | (127 - e) << 24 | 127), 0x38383838, 0x38383838, 32 + 96 * p2(-e));
    return 0;
}
