// Probe: MFMA issue rate of ONE wave per SIMD as a function of the dependent distance (number of accumulators used round-robin), for
// v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16, with 0 / 1 ds_read_b128 per R MFMAs interleaved (results consumed a round later).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mfma_dep_probe.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int BIG, int READS /* one ds_read_b128 after every READS-th MFMA; 0 = none */>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    u32x4* buf = reinterpret_cast<u32x4*>(lds);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    bf16x8 a[4], b = __builtin_bit_cast(bf16x8, buf[lane + 512]);
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, buf[lane + i * 64]);
    float s = 0.f;
    if constexpr (BIG) {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const int nofs = ((it & 7) * 64 + lane);
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k & 3], b, acc[k % NACC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (READS > 0 && k % READS == READS - 1) { a[(k / READS) & 3] = __builtin_bit_cast(bf16x8, buf[nofs + 1024 + (k & 3) * 64]); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    } else {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            const int nofs = ((it & 7) * 64 + lane);
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                acc[k % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k & 3], b, acc[k % NACC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (READS > 0 && k % READS == READS - 1) { a[(k / READS) & 3] = __builtin_bit_cast(bf16x8, buf[nofs + 1024 + (k & 3) * 64]); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// the row-major consumer pattern of conv3_sb2: steps of NM MFMAs on 3 accumulators (product-major), the fragment pair (hi, lo) of step
// s+AHEAD is read right after the first MFMA of step s into a ring of AHEAD+1 slots
template <int NM, int AHEAD>
__global__ __launch_bounds__(512) void probe_steps(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    u32x4* buf = reinterpret_cast<u32x4*>(lds);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 5120; i += blockDim.x) buf[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    constexpr int NS = AHEAD + 1;
    const unsigned long long t_start = __builtin_readcyclecounter();
    bf16x8 fh[NS], fl[NS], w[6];
    for (int i = 0; i < NS; ++i) { fh[i] = __builtin_bit_cast(bf16x8, buf[lane + i * 64]); fl[i] = __builtin_bit_cast(bf16x8, buf[lane + 2176 + i * 64]); }
    for (int i = 0; i < 6; ++i) w[i] = __builtin_bit_cast(bf16x8, buf[lane + 512 + i * 64]);
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
        const int nofs = ((it & 7) * 64 + lane);
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const bf16x8 ah = fh[s % NS], al = fl[s % NS];
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                const int pr = k / (NM / 3), e = k % (NM / 3);
                acc[e % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[(pr == 1 ? 3 : 0) + e % 3], pr == 0 ? al : ah, acc[e % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0) {
                    fh[(s + AHEAD) % NS] = __builtin_bit_cast(bf16x8, buf[nofs + 1024 + s * 18]);
                    __builtin_amdgcn_sched_barrier(0);
                    fl[(s + AHEAD) % NS] = __builtin_bit_cast(bf16x8, buf[nofs + 1024 + 2176 + s * 18]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 3; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sum == 12345.678f) out[threadIdx.x] = sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(out)[512] = __builtin_readcyclecounter() - t_start;   // s_memtime ticks of this wave
}
template <int NM, int AHEAD>
static void run_steps(int threads, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe_steps<NM, AHEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    probe_steps<NM, AHEAD><<<256, threads, 81920>>>(out, 50);
    (void)hipEventRecord(e0);
    probe_steps<NM, AHEAD><<<256, threads, 81920>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wps = threads / 256.0, mf = 12.0 * NM * iters * wps;
    unsigned long long ticks = 0;
    (void)hipMemcpy(&ticks, reinterpret_cast<unsigned long long*>(out) + 512, 8, hipMemcpyDeviceToHost);
    printf("steps of %d MFMAs, fragment pair read %d steps ahead, waves/SIMD %.0f: %6.2f ns per MFMA per SIMD; s_memtime: %.1f ticks per MFMA of one wave, %.3f ticks per ns\n",
           NM, AHEAD, wps, ms * 1e6 / mf, (double)ticks / (12.0 * NM * iters), (double)ticks / (ms * 1e6));
}

template <int NACC, int BIG, int READS>
static void run(int threads, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<NACC, BIG, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    probe<NACC, BIG, READS><<<256, threads, 81920>>>(out, 50);
    hipEventRecord(e0);
    probe<NACC, BIG, READS><<<256, threads, 81920>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wps = threads / 256.0, mf = 24.0 * iters * wps;
    const double ns = ms * 1e6 / mf, flop = BIG ? 32768.0 : 16384.0;
    printf("%s  acc round-robin %d  reads 1/%d  waves/SIMD %.0f: %6.2f ns per MFMA per SIMD -> %5.0f TFLOP/s\n", BIG ? "32x32x16" : "16x16x32", NACC, READS, wps, ns, flop / ns * 1024 / 1e3);
}
int main() {
    float* out; hipMalloc(&out, 1 << 20);
    run<1, 0, 0>(256, out); run<2, 0, 0>(256, out); run<3, 0, 0>(256, out); run<4, 0, 0>(256, out); run<6, 0, 0>(256, out); run<8, 0, 0>(256, out); run<12, 0, 0>(256, out);
    run<3, 0, 3>(256, out); run<4, 0, 3>(256, out); run<6, 0, 3>(256, out); run<8, 0, 3>(256, out); run<6, 0, 2>(256, out); run<8, 0, 2>(256, out); run<8, 0, 1>(256, out);
    run<1, 1, 0>(256, out); run<2, 1, 0>(256, out); run<3, 1, 0>(256, out); run<4, 1, 0>(256, out); run<2, 1, 2>(256, out); run<4, 1, 2>(256, out); run<4, 1, 1>(256, out);
    run_steps<9, 2>(256, out); run_steps<9, 1>(256, out); run_steps<6, 2>(256, out); run_steps<3, 2>(256, out); run_steps<3, 3>(256, out); run_steps<9, 2>(512, out);
    run<4, 0, 0>(512, out); run<4, 0, 3>(512, out); run<8, 0, 3>(512, out); run<2, 1, 0>(512, out); run<4, 1, 2>(512, out);
    return 0;
}
