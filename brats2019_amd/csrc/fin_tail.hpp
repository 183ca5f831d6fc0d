// fin_tail.hpp -- the last workgroup of a launch finalizes the partial sums the launch produced (FinTail in ru_common.h).
//
// Protocol (cdna_hip_programming.md, Guideline 16, the "8-byte agent atomics on both sides" form):
//   every workgroup: partial sums published with stat_publish (ONE aligned 8-byte agent-scope store per (sum, sum2) pair: write-through,
//       never parked in this XCD's L2) -> every wave drains its stores (s_waitcnt vmcnt(0)) -> __syncthreads() -> thread 0 takes a ticket
//       (relaxed agent-scope fetch_add);
//   the workgroup that draws the last ticket reads every pair back with agent-scope loads (served past its own L1 / L2 copies), in a
//       FIXED order, finalizes, and resets the ticket to zero for the next launch (graph replay included).
// The finalized values are written with plain stores: the kernel boundary publishes them to the next kernel.
#pragma once
#include "ru_common.h"

namespace ru {

typedef unsigned long long fin_u64;

__device__ __forceinline__ void stat_publish(float* p, float s1, float s2) {       // p: 8-byte aligned pair
    const fin_u64 v = (fin_u64)__builtin_bit_cast(unsigned, s1) | ((fin_u64)__builtin_bit_cast(unsigned, s2) << 32);
    __hip_atomic_store(reinterpret_cast<fin_u64*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 stat_fetch(const float* p) {
    const fin_u64 v = __hip_atomic_load(reinterpret_cast<const fin_u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__builtin_bit_cast(float, (unsigned)v), __builtin_bit_cast(float, (unsigned)(v >> 32)));
}

// sum of `cnt` consecutive pairs starting at p by the LPI lanes of a lane group (sl = lane % LPI); every lane of the group gets the sums.
// The reads are ONE latency chain (the pairs were stored write-through: they come from the fabric, ~2 us a round trip under load), so a
// lane keeps up to 16 of its loads in flight.  8-byte agent-scope ATOMIC loads, the form Guideline 16 names for both sides (a variant
// reading two pairs per 16-byte sc1 buffer load produced wrong statistics and was dropped: the pair stays the unit of publication AND
// of reading).
template <int LPI>
__device__ __forceinline__ void fin_sum_pairs(const float* p, int cnt, int sl, double& o1, double& o2) {
    double s1 = 0.0, s2 = 0.0;
    for (int i0 = 0; i0 < cnt; i0 += 16 * LPI) {
        float2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = i0 + j * LPI + sl;
            v[j] = i < cnt ? stat_fetch(p + 2 * i) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            s1 += (double)v[j].x + (double)v[j + 1].x;
            s2 += (double)v[j].y + (double)v[j + 1].y;
        }
    }
#pragma unroll
    for (int o = LPI / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    o1 = s1; o2 = s2;
}

// kind 1: one lane group per (sample, group): the cpg*nblk pairs of a group are contiguous
template <int LPI>
__device__ __forceinline__ void fin_gn_stats(const FinTail& f, const float* partials) {
    const int nthreads = blockDim.x, slots = nthreads / LPI, slot = threadIdx.x / LPI, sl = threadIdx.x % LPI;
    const int cpg = f.C / f.G, cnt = cpg * f.nblk;
    const double m = (double)cpg * (double)f.V;
    for (int it = slot; it < f.N * f.G; it += slots) {
        const int n = it / f.G, g = it - n * f.G;
        double s1, s2;
        fin_sum_pairs<LPI>(partials + ((size_t)n * f.C + (size_t)g * cpg) * f.nblk * 2, cnt, sl, s1, s2);
        const double mu = s1 / m;
        double var = s2 / m - mu * mu;
        if (var < 0.0) var = 0.0;
        const double rs = 1.0 / sqrt(var + (double)f.eps);
        const float muf = (float)mu, rsf = (float)rs;
        if (sl == 0) { f.mean[it] = muf; f.rstd[it] = rsf; }
        for (int j = sl; j < cpg; j += LPI) {
            const int c = g * cpg + j;
            const float gm = f.gamma[c], bt = f.beta[c];
            const float a = gm * rsf;
            f.scale[n * f.C + c] = a;
            f.shift[n * f.C + c] = bt - muf * a;
            if (f.bst_k) {                           // constants of the fused GroupNorm-backward statistics (gn_finalize_kernel)
                const float sg = gm < 0.f ? -1.f : 1.f;
                float* kn = f.bst_k + (size_t)n * 3 * f.C;
                kn[c] = sg * rsf;
                kn[f.C + c] = -sg * muf * rsf;
                kn[2 * f.C + c] = gm == 0.f ? (bt > 0.f ? -INFINITY : INFINITY) : -bt / fabsf(gm);
            }
        }
    }
}

// kind 2: phase 1, one lane group per (sample, channel) -> S[n][c][2] in LDS; phase 2, coefficients per (sample, group) and the
// batch-ordered dgamma / dbeta per channel (the arithmetic of gn_bwd_finalize_kernel)
template <int LPI>
__device__ __forceinline__ void fin_gn_bwd_phase1(const FinTail& f, const float* partials, double* S) {
    const int slots = blockDim.x / LPI, slot = threadIdx.x / LPI, sl = threadIdx.x % LPI;
    for (int it = slot; it < f.N * f.C; it += slots) {
        double s1, s2;
        fin_sum_pairs<LPI>(partials + (size_t)it * f.nblk * 2, f.nblk, sl, s1, s2);
        if (f.s2_sign && f.gamma[it % f.C] < 0.f) s2 = -s2;
        if (sl == 0) { S[2 * it] = s1; S[2 * it + 1] = s2; }
    }
}
__device__ __forceinline__ void fin_gn_bwd_phase2(const FinTail& f, const double* S) {
    const int cpg = f.C / f.G;
    const double m = (double)cpg * (double)f.V;
    for (int it = threadIdx.x; it < f.N * f.G; it += blockDim.x) {
        const int n = it / f.G, g = it - n * f.G;
        double m1 = 0.0, m2 = 0.0;
        for (int j = 0; j < cpg; ++j) {
            const double gm = (double)f.gamma[g * cpg + j];
            m1 += gm * S[(n * f.C + g * cpg + j) * 2];
            m2 += gm * S[(n * f.C + g * cpg + j) * 2 + 1];
        }
        m1 /= m; m2 /= m;
        const double mu = (double)f.mean[it], rs = (double)f.rstd[it];
        for (int j = 0; j < cpg; ++j) {
            const int c = g * cpg + j;
            float* q = f.coef + ((size_t)n * f.C + c) * 3;
            q[0] = (float)(rs * (double)f.gamma[c]);
            q[1] = (float)(-rs * rs * m2);
            q[2] = (float)(rs * rs * m2 * mu - rs * m1);
        }
    }
    for (int c = threadIdx.x; c < f.C; c += blockDim.x) {
        double dg = 0.0, db = 0.0;
        for (int n = 0; n < f.N; ++n) { db += S[(n * f.C + c) * 2]; dg += S[(n * f.C + c) * 2 + 1]; }
        if (f.dgamma) f.dgamma[c] = (float)dg;
        if (f.dbeta) f.dbeta[c] = (float)db;
    }
}

// Called by EVERY thread of EVERY workgroup of the launch, after the workgroup's last stat_publish, from uniform control flow.
// lds: LDS the workgroup no longer needs -- 8 bytes, plus fin_tail_lds_bytes() for kind 2 (16-byte aligned).
__device__ __forceinline__ void fin_tail(const FinTail& f, const float* partials, void* lds) {
    if (!f.ticket) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's published pairs have left the CU
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(lds);
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        // acq_rel at agent scope: the arrival releases this workgroup's published partials (L2 write-back) and the last arriver's read of
        // the others' is ordered behind its own ticket (L1 invalidate) -- no reliance on the publish / fetch forms alone
        const unsigned t = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        *flag = (t == total - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u) return;                                     // (workgroup-uniform)
    if (f.kind == 1) {
        const int cnt = (f.C / f.G) * f.nblk;                    // pairs per item: one pass of <= 16 pairs per lane where the lanes allow
        if (cnt > 256) fin_gn_stats<64>(f, partials);
        else if (cnt > 64) fin_gn_stats<16>(f, partials);
        else fin_gn_stats<4>(f, partials);
    } else {
        double* S = reinterpret_cast<double*>(reinterpret_cast<char*>(lds) + 16);
        if (f.nblk > 256) fin_gn_bwd_phase1<64>(f, partials, S);
        else if (f.nblk > 64) fin_gn_bwd_phase1<16>(f, partials, S);
        else fin_gn_bwd_phase1<4>(f, partials, S);
        __syncthreads();
        fin_gn_bwd_phase2(f, S);
    }
    if (threadIdx.x == 0) __hip_atomic_store(f.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace ru
