// Probe (round-5 verdict, item 1b): what does the MATRIX PIPE pay for one 27-tap x 16-channel product chain (K = 432) in the two product schemes,
// on random operands, in loops long enough for the socket power limit to settle?
//   today:      3 x 14 v_mfma_f32_16x16x32_bf16                           (hi*hi + lo*hi + hi*lo; the ninth tap pair is half phantom)
//   candidate:  14 v_mfma_f32_16x16x32_f16 + 2 x 4 v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, constant E8M0 scales; 32 tap slots for 27 taps)
// plus each instruction class alone.  One or two waves per SIMD on every CU, three accumulators round-robin (the conv consumer's pattern), operands from
// registers (no LDS traffic: this prices the products, not the kernel).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 42 bf16   1: 14 f16 + 8 mx-f8   2: 14 bf16 (one product)   3: 8 mx-f8 alone   4: 14 f16 alone   5: 28 bf16 (two products)
template <int MODE>
__global__ __launch_bounds__(512) void probe(const u32x4* __restrict__ src, float* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 r[12];
    for (int i = 0; i < 12; ++i) r[i] = src[(blockIdx.x * 8 + wave) * 64 * 12 + i * 64 + lane];
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const int sc = 0x7f7f7f7f;      // E8M0 127 = 2^0 in every byte
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 2 || MODE == 5) {
            constexpr int N = MODE == 0 ? 42 : (MODE == 2 ? 14 : 28);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                acc[k % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, r[k % 6]), __builtin_bit_cast(bf16x8, r[6 + k % 5]), acc[k % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (MODE == 1 || MODE == 4) {
#pragma unroll
            for (int k = 0; k < 14; ++k) {
                acc[k % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, r[k % 6]), __builtin_bit_cast(f16x8, r[6 + k % 5]), acc[k % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                i32x8 a, b;
                const u32x4 a0 = r[(2 * k) % 6], a1 = r[(2 * k + 1) % 6], b0 = r[6 + k % 5], b1 = r[6 + (k + 1) % 5];
                a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3]; a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
                b[0] = b0[0]; b[1] = b0[1]; b[2] = b0[2]; b[3] = b0[3]; b[4] = b1[0]; b[5] = b1[1]; b[6] = b1[2]; b[7] = b1[3];
                acc[k % 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[k % 3], 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, sc, 0, sc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // keep the accumulators bounded (random operands would run them to inf, which changes the data-dependent power): one cheap scale per chain
        acc[0] *= 0.5f; acc[1] *= 0.5f; acc[2] *= 0.5f;
    }
    float s = 0.f;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

template <int MODE>
static double run(const char* what, double units, int threads, const u32x4* src, float* out, int iters, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE><<<256, threads>>>(src, out, iters / 10);
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) probe<MODE><<<256, threads>>>(src, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wps = threads / 256.0, chains = (double)iters * reps * wps;     // chains per SIMD
    const double ns = ms * 1e6 / chains;
    printf("%-58s waves/SIMD %.0f: %8.1f ns per K=432 chain per SIMD (%5.2f ns per bf16-MFMA-equivalent unit; %d x %.1f ms)\n", what, wps, ns, ns / units, reps, ms / reps);
    return ns;
}

int main() {
    // operands: a 16-byte register holds 8 x 16-bit or 16 x 8-bit values.  The same random bits are read as bf16 / fp16 / e4m3 by the three instruction
    // classes; exponents are kept small so that no class sees inf / NaN: 16-bit patterns with the top exponent bits cleared, bytes never 0x7f / 0xff.
    const size_t n = (size_t)256 * 8 * 64 * 12;
    std::vector<uint32_t> h(n * 4);
    for (auto& v : h) {
        uint32_t w = 0;
        for (int b = 0; b < 4; ++b) {
            uint32_t byte = rnd() & 0xff;
            if ((b & 1) == 1) byte &= 0xbf;           // high byte of a 16-bit value: clear the top exponent bit (bf16: |v| < 2; fp16: |v| < 2)
            if ((byte & 0x7f) == 0x7f) byte ^= 0x01;  // e4m3 NaN
            w |= byte << (8 * b);
        }
        v = w;
    }
    u32x4* src; float* out;
    (void)hipMalloc(&src, n * 16); (void)hipMalloc(&out, 1 << 20);
    (void)hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    const int iters = 20000, reps = 20;
    for (int threads : {256, 512}) {
        const double t0 = run<0>("3 x 14 bf16 16x16x32 (today)", 42, threads, src, out, iters, reps);
        const double t1 = run<1>("14 f16 16x16x32 + 8 mx-e4m3 16x16x128 (candidate)", 30, threads, src, out, iters, reps);
        run<2>("14 bf16 16x16x32 (one product)", 14, threads, src, out, iters, reps);
        run<5>("28 bf16 16x16x32 (two products)", 28, threads, src, out, iters, reps);
        run<4>("14 f16 16x16x32", 14, threads, src, out, iters, reps);
        run<3>("8 mx-e4m3 16x16x128", 16, threads, src, out, iters, reps);
        // again, interleaved, so that a thermal / clock drift between the first two lines shows
        const double t0b = run<0>("3 x 14 bf16 16x16x32 (today), again", 42, threads, src, out, iters, reps);
        const double t1b = run<1>("candidate, again", 30, threads, src, out, iters, reps);
        printf("   -> today / candidate = %.3f, %.3f (the gate: >= 1.3)\n", t0 / t1, t0b / t1b);
    }
    return 0;
}
