// Probe: (a) v_fma_mix_f32 with an fp16 source half computes t - f16(t) exactly like cvt + sub; (b) v_cvt_scalef32_pk_fp8_f32(x, scale = 2^-11) under MODE.FP16_OVFL
// equals v_cvt_pk_fp8_f32(x * 2^11) byte for byte, saturation and subnormals included (so the staging can drop its multiply).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, int n, unsigned* bad) {
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    h2 h; h[0] = (_Float16)a; h[1] = (_Float16)b;
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    const float la = a - (float)h[0], lb = b - (float)h[1];
    float ma, mb;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ma) : "v"(hb), "v"(a));
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(mb) : "v"(hb), "v"(b));
    if (__builtin_bit_cast(unsigned, la) != __builtin_bit_cast(unsigned, ma) || __builtin_bit_cast(unsigned, lb) != __builtin_bit_cast(unsigned, mb)) atomicAdd(&bad[0], 1u);
    const unsigned p = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(la * 2048.f, lb * 2048.f, 0, false) & 0xffffu;
    const s16x2 w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(s16x2{0, 0}, la, lb, 1.f / 2048.f, false);
    if (p != (unsigned)(unsigned short)w[0]) atomicAdd(&bad[1], 1u);
    // the same on the values themselves (wide range: saturation at 448 * 2^-11, subnormals)
    const unsigned p2 = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a * 2048.f, b * 2048.f, 0, false) & 0xffffu;
    const s16x2 w2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(s16x2{0, 0}, a, b, 1.f / 2048.f, false);
    if (p2 != (unsigned)(unsigned short)w2[0]) { if (atomicAdd(&bad[2], 1u) < 8) printf("  x = %g, %g: mul + cvt 0x%04x, scaled cvt 0x%04x\n", a, b, p2, (unsigned)(unsigned short)w2[0]); }
}
int main() {
    const int n = 1 << 20;
    std::vector<float> xs(n);
    uint32_t st = 99u;
    for (auto& v : xs) { st = st * 1664525u + 1013904223u; const float u = (float)(st >> 8) / 16777216.f; v = ldexpf(u * 2.f - 1.f, (int)((st >> 3) & 31) - 24); }
    float* dx; unsigned* db; hipMalloc(&dx, n * 4); hipMalloc(&db, 16); hipMemset(db, 0, 16);
    hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(dx, n, db);
    unsigned bad[4]; hipMemcpy(bad, db, 16, hipMemcpyDeviceToHost);
    printf("%d values in 2^-24 .. 2^7: fma_mix lo != cvt + sub in %u pairs; scaled cvt != mul + cvt on the residuals in %u pairs, on the values in %u pairs\n", n, bad[0], bad[1], bad[2]);
    return 0;
}
