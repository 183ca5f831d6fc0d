// Per-LANE scales of v_mfma_scale_f32_16x16x128_f8f6f4 (round 6, section 12 of profiles/r06_notes.txt): does block b of column n take its B scale from byte 0 of lane (n, group b)?
// A = B = e4m3 1.0 everywhere, A scale 2^0; B scale byte 0 of lane (n, g) = 127 - (n % 8) - 8 g, bytes 1-3 = 2^-60.  Expected D[m][n] = 32 * sum_b 2^-((n % 8) + 8 b).
// The same for the A operand (rows m).   hipcc --offload-arch=gfx950 -O2 tools/mx_scale_probe3.hip -o /tmp/mx_scale_probe3 && /tmp/mx_scale_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, int which) {
    const int lane = threadIdx.x, g = lane >> 4, n = lane & 15;
    i32x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = 0x38383838; bv[j] = 0x38383838; }
    const int tiny = 127 - 60;
    const int mine = (127 - (n % 8) - 8 * g) | (tiny << 8) | (tiny << 16) | (tiny << 24);
    const int one = 0x7f7f7f7f;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (which == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, one, 0, mine);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, mine, 0, one);
    for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + n] = c[r];       // D[m = 4 g + r][n]
}
int main() {
    float* d; hipMalloc(&d, 1024);
    for (int which = 0; which < 2; ++which) {
        k<<<1, 64>>>(d, which);
        float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                const int idx = which == 0 ? n : m;
                double e = 0; for (int b = 0; b < 4; ++b) e += 32.0 * std::ldexp(1.0, -((idx % 8) + 8 * b));
                if (std::fabs(h[m * 16 + n] - e) > 1e-6 * e) { if (bad < 4) printf("  D[%d][%d] = %.9g expected %.9g\n", m, n, h[m * 16 + n], e); ++bad; }
            }
        printf("per-lane %s scales (byte 0 of lane (index, group b) scales block b of that %s): %d of 256 outputs differ from the model\n",
               which == 0 ? "B" : "A", which == 0 ? "column" : "row", bad);
    }
    return 0;
}
