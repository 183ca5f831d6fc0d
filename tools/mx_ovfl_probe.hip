// Probe: does MODE.FP16_OVFL (hwreg MODE bit 23) make v_cvt_pk_fp8_f32 / v_cvt_pk_f16_f32 SATURATE instead of producing NaN / inf?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* o, int set) {
    if (set) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    const int i = threadIdx.x;
    const float a = x[2 * i], b = x[2 * i + 1];
    o[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
    h2 h; h[0] = (_Float16)a; h[1] = (_Float16)b;
    o[64 + i] = __builtin_bit_cast(unsigned, h);
    o[128 + i] = __builtin_bit_cast(unsigned, a * 3.0f);      // plain f32 arithmetic must be unaffected
}
int main() {
    float hx[16] = {1.f, 447.f, 464.f, 480.f, 500.f, 1e4f, -1e6f, 3e38f, 65504.f, 65520.f, 7e4f, -1e5f, 1e-3f, 0.3f, __builtin_inff(), -__builtin_inff()};
    float* dx; unsigned* d; hipMalloc(&dx, 64); hipMalloc(&d, 192 * 4);
    hipMemcpy(dx, hx, 64, hipMemcpyHostToDevice);
    for (int set = 0; set < 2; ++set) {
        k<<<1, 8>>>(dx, d, set);
        unsigned h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("FP16_OVFL = %d\n", set);
        for (int i = 0; i < 16; ++i)
            printf("  x = %-12g e4m3 0x%02x   f16 0x%04x   (3x as f32: %g)\n", hx[i], (h[i / 2] >> (8 * (i & 1))) & 0xff, (h[64 + i / 2] >> (16 * (i & 1))) & 0xffff, i % 2 == 0 ? __builtin_bit_cast(float, h[128 + i / 2]) : 0.f);
    }
    return 0;
}
