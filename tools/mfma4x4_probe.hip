// Which lane's A / B element feeds D[vgpr r] of lane l in v_mfma_f32_4x4x1_16b_f32?  (round 5: the exact-f32 head conv has 3 output channels --
// a 16-wide N tile wastes 13/16 of the matrix work, sixteen 4x4 blocks waste 1/4.)   hipcc --offload-arch=gfx950 tools/mfma4x4_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int mode) {
    const int l = threadIdx.x;
    const float a = mode == 0 ? (float)(l + 1) : 1.f;
    const float b = mode == 0 ? 1.f : (float)(l + 1);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * sizeof(float));
    float h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s operand: D[lane][vgpr] = 1 + index of the contributing lane\n", mode == 0 ? "A" : "B");
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int r = 0; r < 4; ++r) printf(" %3.0f", h[l * 4 + r] - 1); printf("%s", (l & 3) == 3 ? "\n" : "   |  "); }
    }
    return 0;
}
