// Probe: what does a VALU / LDS-write stream in a SECOND wave of the same SIMD cost the wave that issues the MFMAs?
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) issue v_mfma_f32_16x16x32_bf16 back to back on 4 accumulators, waves
// 4-7 (the other wave of each SIMD) run a stream of ONE kind of instruction until the MFMA waves are done and count how many they got
// through.  Printed: ns per MFMA next to each stream, and the stream's own rate (instructions per microsecond and wave).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { K_IDLE = 0, K_FMA, K_PK_FMA, K_PK_ADD, K_PK_MUL, K_CVT, K_CNDMASK, K_SHIFT, K_MAX, K_SUB, K_DSW, K_DSR, K_NKIND };
static const char* kind_name[K_NKIND] = {"(none)", "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_cvt_pk_bf16_f32", "v_cndmask_b32",
                                          "v_lshlrev_b32", "v_max_f32", "v_sub_f32", "ds_write_b128", "ds_read_b128"};

template <int KIND>
__device__ __forceinline__ void stream16(f32x2 (&r)[8], float c, u32x4* buf, int lane) {       // 16 instructions on 8 independent registers pairs
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i][0]) : "v"(c));
            else if constexpr (KIND == K_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(f32x2{c, c}));
            else if constexpr (KIND == K_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(f32x2{c, c}));
            else if constexpr (KIND == K_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(f32x2{c, c}));
            else if constexpr (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r[i][0]) : "v"(r[i][1]), "v"(c));
            else if constexpr (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i][0]) : "v"(c) : "vcc");
            else if constexpr (KIND == K_SHIFT) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i][0]));
            else if constexpr (KIND == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i][0]) : "v"(c));
            else if constexpr (KIND == K_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i][0]) : "v"(c));
            else if constexpr (KIND == K_DSW) { if (i < 2) buf[2048 + lane + 64 * i + 128 * rep] = u32x4{__builtin_bit_cast(unsigned, r[i][0]), 1u, 2u, 3u}; }
            else if constexpr (KIND == K_DSR) { if (i < 2) { const u32x4 v = buf[2048 + lane + 64 * i + 128 * rep]; r[i][0] += __builtin_bit_cast(float, v[0]); } }
        }
    }
}

template <int KIND, bool WITH_MFMA>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* stat, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    u32x4* buf = reinterpret_cast<u32x4*>(lds);
    volatile int* done = reinterpret_cast<volatile int*>(lds + 65536);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    if (threadIdx.x == 0) *done = 0;
    __syncthreads();
    if (wave < 4) {
        if (!WITH_MFMA) return;
        bf16x8 a[4], b = __builtin_bit_cast(bf16x8, buf[lane + 512]);
        for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, buf[lane + i * 64]);
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k & 3], b, acc[k & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678f) out[threadIdx.x] = s;
        if (lane == 0) atomicAdd(const_cast<int*>(done), 1);
        if (blockIdx.x == 0 && threadIdx.x == 0) stat[0] = t1 - t0;
    } else {
        if (KIND == K_IDLE) return;
        f32x2 r[8];
        for (int i = 0; i < 8; ++i) r[i] = f32x2{1.f + lane * 1e-3f, 0.5f};
        const float c = 1.0000001f;
        unsigned long long n = 0;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; WITH_MFMA ? *done < 4 : it < iters; ++it) {
#pragma unroll 1
            for (int q = 0; q < 16; ++q) stream16<KIND>(r, c, buf, lane);
            n += 256;
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += r[i][0] + r[i][1];
        if (s == 12345.678f) out[threadIdx.x] = s;
        if (blockIdx.x == 0 && threadIdx.x == 256) { stat[1] = n; stat[2] = t1 - t0; }
    }
}

template <int KIND, bool WITH_MFMA>
static void run(float* out, unsigned long long* stat) {
    const int iters = WITH_MFMA ? 20000 : 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<KIND, WITH_MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    (void)hipMemset(stat, 0, 64);
    probe<KIND, WITH_MFMA><<<256, 512, 81920>>>(out, stat, 200);
    (void)hipEventRecord(e0);
    probe<KIND, WITH_MFMA><<<256, 512, 81920>>>(out, stat, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[3] = {0, 0, 0};
    (void)hipMemcpy(h, stat, 24, hipMemcpyDeviceToHost);
    const double nm = 24.0 * iters;
    if (WITH_MFMA) printf("%-20s beside MFMAs: kernel %7.3f ms = %5.2f ns per MFMA | the stream: %6.2f ns per instruction\n", kind_name[KIND], ms, ms * 1e6 / nm,
                          h[1] ? ms * 1e6 / h[1] : 0.0);
    else printf("%-20s alone:                                               | the stream: %6.2f ns per instruction\n", kind_name[KIND], h[1] ? ms * 1e6 / h[1] : 0.0);
}

// the same question INSIDE the MFMA wave: after every third MFMA, one packed-f32 instruction or the two plain ones it replaces
template <int MIX /* 0 none, 1 v_pk_add_f32, 2 two v_add_f32, 3 v_pk_fma_f32, 4 two v_fma_f32 */>
__global__ __launch_bounds__(256) void probe_mix(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    u32x4* buf = reinterpret_cast<u32x4*>(lds);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    bf16x8 a[4], b = __builtin_bit_cast(bf16x8, buf[lane + 512]);
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, buf[lane + i * 64]);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x2 r[8];
    for (int i = 0; i < 8; ++i) r[i] = f32x2{1.f + lane * 1e-3f, 0.5f};
    const float c = 1.0000001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k & 3], b, acc[k & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k % 3 == 2) {
                const int i = (k / 3) & 7;
                if constexpr (MIX == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(f32x2{c, c}));
                else if constexpr (MIX == 2) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i][0]) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i][1]) : "v"(c)); }
                else if constexpr (MIX == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(f32x2{c, c}));
                else if constexpr (MIX == 4) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i][0]) : "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i][1]) : "v"(c)); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += r[i][0] + r[i][1];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int MIX>
static void run_mix(float* out) {
    const int iters = 20000;
    static const char* nm[5] = {"nothing", "one v_pk_add_f32", "two v_add_f32", "one v_pk_fma_f32", "two v_fma_f32"};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe_mix<MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    probe_mix<MIX><<<256, 256, 81920>>>(out, 200);
    (void)hipEventRecord(e0);
    probe_mix<MIX><<<256, 256, 81920>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("MFMA wave with %-18s after every third MFMA: %5.2f ns per MFMA\n", nm[MIX], ms * 1e6 / (24.0 * iters));
}

// MFMA only, one wave per SIMD, operands either all 1.0 (the probes above) or pseudo-random bf16 in (-2, 2): the matrix pipe's power
// depends on the data, and the chip runs into its power cap with real data long before the issue rate is the limit
template <int RANDOM, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe_data(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    u32x4* buf = reinterpret_cast<u32x4*>(lds);
    const int lane = threadIdx.x & 63;
    unsigned st = 0x9e3779b9u * (threadIdx.x + 1) + blockIdx.x * 7919u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; const unsigned m = (st >> 9) & 0x7fu, e = 0x3f00u + ((st >> 20) & 0x80u), sg = (st >> 16) & 0x8000u; return (sg | e | m) & 0xffffu; };
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) {
        u32x4 v;
        for (int j = 0; j < 4; ++j) v[j] = RANDOM ? (rnd() | (rnd() << 16)) : 0x3f803f80u;
        buf[i] = v;
    }
    __syncthreads();
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = __builtin_bit_cast(bf16x8, buf[lane + i * 64]); b[i] = __builtin_bit_cast(bf16x8, buf[lane + 512 + i * 64]); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k & 3], b[(k >> 2) & 3], acc[k & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int RANDOM, int WAVES>
static void run_data(float* out, int launches) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe_data<RANDOM, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    probe_data<RANDOM, WAVES><<<256, WAVES * 64, 81920>>>(out, 200);
    (void)hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) probe_data<RANDOM, WAVES><<<256, WAVES * 64, 81920>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (24.0 * iters * launches * (WAVES / 4.0));
    printf("MFMA only, %d wave(s) per SIMD, %s operands, %3d launches of %.1f ms: %5.2f ns per MFMA per SIMD -> %5.0f TFLOP/s\n", WAVES / 4, RANDOM ? "random" : "all-ones",
           launches, ms / launches, ns, 16384.0 / ns * 1024 / 1e3);
}

int main() {
    float* out; (void)hipMalloc(&out, 1 << 20);
    unsigned long long* stat; (void)hipMalloc(&stat, 64);
    for (int rep = 0; rep < 2; ++rep) {
#define RU_BOTH(K) run<K, true>(out, stat); run<K, false>(out, stat)
        run<K_IDLE, true>(out, stat);
        RU_BOTH(K_FMA); RU_BOTH(K_PK_FMA); RU_BOTH(K_PK_ADD); RU_BOTH(K_PK_MUL); RU_BOTH(K_CVT); RU_BOTH(K_CNDMASK); RU_BOTH(K_SHIFT); RU_BOTH(K_MAX);
        RU_BOTH(K_SUB); RU_BOTH(K_DSW); RU_BOTH(K_DSR);
        run<K_IDLE, true>(out, stat);
    }
    run_data<0, 4>(out, 1); run_data<1, 4>(out, 1); run_data<0, 4>(out, 50); run_data<1, 4>(out, 50); run_data<0, 8>(out, 50); run_data<1, 8>(out, 50);
    for (int rep = 0; rep < 2; ++rep) { run_mix<0>(out); run_mix<1>(out); run_mix<2>(out); run_mix<3>(out); run_mix<4>(out); }
    return 0;
}
