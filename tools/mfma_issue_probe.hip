// Probe: how many MFMAs per second does ONE wave per SIMD sustain when LDS fragment reads sit between its MFMAs, and what does a
// second wave on the SIMD change?  The loop mimics the consumer of conv3_sb2: 4 accumulators, 12 MFMAs + 8 ds_read_b128 per "K-step".
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mfma_issue_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMAs only; 1: + 8 reads per 12 MFMAs (results used by the next step's MFMAs); 2: reads but results unused (no waits)
// MODE 3: as 1 + 16 independent VALU FMAs per K-step; MODE 4: as 1 + one global_store_dwordx4 (1 KB per wave) every 3rd K-step;
// MODE 5: as 4 with 16 VALU ops in front of every store (an epilogue row)
template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    u32x4* buf = reinterpret_cast<u32x4*>(lds);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 ah[4], al[4];
    for (int i = 0; i < 4; ++i) { ah[i] = __builtin_bit_cast(bf16x8, buf[lane + i * 64]); al[i] = __builtin_bit_cast(bf16x8, buf[lane + 256 + i * 64]); }
    const bf16x8 bh = __builtin_bit_cast(bf16x8, buf[lane + 512]), bl = __builtin_bit_cast(bf16x8, buf[lane + 576]);
    u32x4 sink = {0, 0, 0, 0};
    constexpr int RM = (MODE >= 3) ? 1 : MODE;           // read behaviour
    float vf[16];
    for (int i = 0; i < 16; ++i) vf[i] = (float)(lane + i);
    float4* gout = reinterpret_cast<float4*>(out) + 4096 + (size_t)(blockIdx.x * blockDim.x + threadIdx.x);
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (int it = 0; it < iters; ++it) {
        const int nofs = ((it & 7) * 64 + lane);
        if (MODE == 3 || (MODE == 5 && it % 3 == 2)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) vf[i] = __builtin_fmaf(vf[i], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((MODE == 4 || MODE == 5) && it % 3 == 2) {
            gout[(size_t)(it & 63) * gstride] = make_float4(acc[0][0] + vf[0], acc[1][1] + vf[1], acc[2][2] + vf[2], acc[3][3] + vf[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (RM == 1 && i > 0) { al[i - 1] = __builtin_bit_cast(bf16x8, buf[nofs + 1024 + (i - 1) * 64]); __builtin_amdgcn_sched_barrier(0); }
            if (MODE == 2 && i > 0) { const u32x4 t = buf[nofs + 1024 + (i - 1) * 64]; sink[0] ^= t[0]; __builtin_amdgcn_sched_barrier(0); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (RM == 1 && i == 0) { al[3] = __builtin_bit_cast(bf16x8, buf[nofs + 1024 + 3 * 64]); __builtin_amdgcn_sched_barrier(0); }
            if (MODE == 2 && i == 0) { const u32x4 t = buf[nofs + 1024 + 3 * 64]; sink[1] ^= t[0]; __builtin_amdgcn_sched_barrier(0); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (RM == 1) { ah[i] = __builtin_bit_cast(bf16x8, buf[nofs + 2048 + i * 64]); __builtin_amdgcn_sched_barrier(0); }
            if (MODE == 2) { const u32x4 t = buf[nofs + 2048 + i * 64]; sink[2] ^= t[0]; __builtin_amdgcn_sched_barrier(0); }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += vf[i];
    if (s == 12345.678f || sink[0] + sink[1] + sink[2] == 0x12345u) out[threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int threads, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    probe<MODE><<<256, threads, 81920>>>(out, 100);            // LDS 80 KB: one workgroup per CU
    hipEventRecord(e0);
    probe<MODE><<<256, threads, 81920>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 256.0, mf = 12.0 * iters * waves_per_simd;      // MFMAs per SIMD
    const double ns = ms * 1e6 / mf;
    printf("%-34s waves/SIMD %.0f: %.2f ns per MFMA per SIMD  -> %.0f TFLOP/s (256 CUs)\n", name, waves_per_simd, ns, 16384.0 / ns * 1024 / 1e3);
}
int main() {
    float* out; hipMalloc(&out, (size_t)256 << 20);
    for (int t : {256, 512, 768}) {
        if (t == 256) { run<0>("MFMA only", 256, out); run<1>("MFMA + 8 reads/12 (used)", 256, out); run<2>("MFMA + 8 reads/12 (unused)", 256, out);
                        run<3>("reads + 16 VALU per K-step", 256, out); run<4>("reads + 1 store per 3 K-steps", 256, out); run<5>("reads + (16 VALU, store) per 3", 256, out); }
        if (t == 512) { run<3>("reads + 16 VALU per K-step", 512, out); run<4>("reads + 1 store per 3 K-steps", 512, out); run<0>("MFMA only", 512, out); run<1>("MFMA + 8 reads/12 (used)", 512, out); run<2>("MFMA + 8 reads/12 (unused)", 512, out); }
        if (t == 768) { run<0>("MFMA only", 768, out); run<1>("MFMA + 8 reads/12 (used)", 768, out); }
    }
    return 0;
}
