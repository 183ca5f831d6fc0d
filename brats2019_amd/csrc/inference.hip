// inference.hip -- the device side of the reference's per-case inference driver: everything between the upload of a
// case and the download of its label volume that is not the network itself.
//
//   test.py:47-49,85-87     bounding box of the non-zero voxels           -> bbox_kernel            (ru_case_bbox)
//   test.py:92-113          crop, zero-pad to x16, non-zero z-score       -> case_stats_*_kernel + case_prepare_kernel
//   test.py:115-120         the four test-time flips                      -> written by case_prepare_kernel as one batch
//   test.py:134-144         un-flip, average, un-pad, threshold, count    -> tta_merge_box_kernel   (ru_tta_merge_box)
//   test.py:51-62,162-164   26-connected components, ratio-0.1 rejection  -> cc_* kernels           (ru_cc_reject)
//   test.py:167-168         paste into the full volume                    -> paste_labels_kernel    (ru_paste_labels)
//   loader_helper.py:42-60  zero-padded tile extract (`copy`)             -> tile_gather_kernel     (ru_tile_gather), T tiles per launch
//   loader_helper.py:82-97  centre paste (`copy_back`)                    -> tile_scatter_kernel    (ru_tile_scatter)
//
// All of it is byte / index work bound by HBM: coalesced rows, integer atomics only (results do not depend on the order of execution),
// float64 sums in a fixed two-stage order.
#include "ru_common.h"
#include "pw_helpers.hpp"

#include <limits.h>

namespace ru {

struct Box3 { int lo[3], size[3]; };

// ------------------------------------------------------------------ sliding-window tiles (loader_helper.py:34-97, train.py:158-174)
constexpr int kMaxTiles = 64;
struct TileOrigins { int o[kMaxTiles][3]; };

// tiles[(t * N + n), c, z, y, x] = data[n, c, oz + z, oy + y, ox + x] or 0 outside the volume; a thread writes 4 consecutive x
__global__ __launch_bounds__(256) void tile_gather_kernel(const float* __restrict__ data, float* __restrict__ tiles, const TileOrigins org, int N, int C,
                                                          int D, int H, int W, int td, int th, int tw) {
    const int tnc = blockIdx.y;                            // (t * N + n) * C + c
    const int c = tnc % C, n = (tnc / C) % N, t = tnc / (C * N);
    const int oz = org.o[t][0], oy = org.o[t][1], ox = org.o[t][2];
    const float* src = data + ((size_t)n * C + c) * D * H * W;
    float4* dst = reinterpret_cast<float4*>(tiles + (size_t)tnc * td * th * tw);
    const int tw4 = tw >> 2;
    const size_t total = (size_t)td * th * tw4;
    for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < total; f += (size_t)gridDim.x * 256) {
        const int x4 = (int)(f % tw4);
        const size_t r = f / tw4;
        const int y = (int)(r % th), z = (int)(r / th);
        const int sz = oz + z, sy = oy + y, sx = ox + 4 * x4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sz >= 0 && sz < D && sy >= 0 && sy < H) {
            const float* row = src + ((size_t)sz * H + sy) * W;
            v.x = (sx >= 0 && sx < W) ? row[sx] : 0.f;
            v.y = (sx + 1 >= 0 && sx + 1 < W) ? row[sx + 1] : 0.f;
            v.z = (sx + 2 >= 0 && sx + 2 < W) ? row[sx + 2] : 0.f;
            v.w = (sx + 3 >= 0 && sx + 3 < W) ? row[sx + 3] : 0.f;
        }
        dst[f] = v;
    }
}

// out[n, c, lo + z, ...] = tiles[(t * N + n), c, b + z, ...] for the centre block of tile t, clipped at the volume end
__global__ __launch_bounds__(256) void tile_scatter_kernel(const float* __restrict__ tiles, float* __restrict__ out, const TileOrigins org, int N, int C,
                                                           int D, int H, int W, int td, int th, int tw, int bz, int by, int bx, int cd, int ch, int cw) {
    const int tnc = blockIdx.y;
    const int c = tnc % C, n = (tnc / C) % N, t = tnc / (C * N);
    const int lz = org.o[t][0] + bz, ly = org.o[t][1] + by, lx = org.o[t][2] + bx;     // first voxel of the centre block in the volume (>= 0)
    const float* src = tiles + (size_t)tnc * td * th * tw;
    float* dst = out + ((size_t)n * C + c) * D * H * W;
    const size_t total = (size_t)cd * ch * cw;
    for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < total; f += (size_t)gridDim.x * 256) {
        const int x = (int)(f % cw);
        const size_t r = f / cw;
        const int y = (int)(r % ch), z = (int)(r / ch);
        if (lz + z < D && ly + y < H && lx + x < W)
            dst[((size_t)(lz + z) * H + (ly + y)) * W + lx + x] = src[((size_t)(bz + z) * th + (by + y)) * tw + bx + x];
    }
}

// ------------------------------------------------------------------ bounding box (test.py:47-49; loader_helper.py:105-129)
// box[c] = {min z, min y, min x, max z, max y, max x} over the non-zero voxels of modality c (INT_MAX / -1 when it has none): integer
// atomics, one per block and bound
__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ image, int* __restrict__ box, int D, int H, int W) {
    __shared__ int sm[6];
    const int c = blockIdx.y;
    if (threadIdx.x < 3) sm[threadIdx.x] = INT_MAX;
    else if (threadIdx.x < 6) sm[threadIdx.x] = -1;
    __syncthreads();
    const size_t V = (size_t)D * H * W;
    const float* p = image + (size_t)c * V;
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {-1, -1, -1};
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        if (p[v] != 0.f) {
            const int x = (int)(v % W);
            const size_t r = v / W;
            const int y = (int)(r % H), z = (int)(r / H);
            lo[0] = min(lo[0], z); lo[1] = min(lo[1], y); lo[2] = min(lo[2], x);
            hi[0] = max(hi[0], z); hi[1] = max(hi[1], y); hi[2] = max(hi[2], x);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (lo[a] != INT_MAX) atomicMin(&sm[a], lo[a]);
        if (hi[a] >= 0) atomicMax(&sm[3 + a], hi[a]);
    }
    __syncthreads();
    if (threadIdx.x < 3) { if (sm[threadIdx.x] != INT_MAX) atomicMin(&box[c * 6 + threadIdx.x], sm[threadIdx.x]); }
    else if (threadIdx.x < 6) { if (sm[threadIdx.x] >= 0) atomicMax(&box[c * 6 + threadIdx.x], sm[threadIdx.x]); }
}
__global__ void bbox_init_kernel(int* box, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C * 6) box[i] = (i % 6) < 3 ? INT_MAX : -1;
}

// ------------------------------------------------------------------ z-score of the crop (test.py:103-113)
// per channel over the crop box: count(x > 0), sum x, sum x^2 in float64 (the reference divides a float32 array by an int64 count: its
// moments are float64); partials per 16384-voxel chunk, combined in chunk order
constexpr int CS_CHUNK = 16384;
__global__ __launch_bounds__(256) void case_stats_partial_kernel(const float* __restrict__ image, double* __restrict__ part, int D, int H, int W, Box3 b, int nblk) {
    __shared__ double buf[4];
    const int c = blockIdx.y;
    const size_t Vb = (size_t)b.size[0] * b.size[1] * b.size[2];
    const size_t v0 = (size_t)blockIdx.x * CS_CHUNK, v1 = v0 + CS_CHUNK < Vb ? v0 + CS_CHUNK : Vb;
    const float* p = image + (size_t)c * D * H * W;
    double n = 0.0, s1 = 0.0, s2 = 0.0;
    for (size_t v = v0 + threadIdx.x; v < v1; v += 256) {
        const int x = (int)(v % b.size[2]);
        const size_t r = v / b.size[2];
        const int y = (int)(r % b.size[1]), z = (int)(r / b.size[1]);
        const double t = (double)p[((size_t)(b.lo[0] + z) * H + (b.lo[1] + y)) * W + b.lo[2] + x];
        n += t > 0.0 ? 1.0 : 0.0;
        s1 += t;
        s2 += t * t;
    }
    n = block_sum_d(n, buf);
    s1 = block_sum_d(s1, buf);
    s2 = block_sum_d(s2, buf);
    if (threadIdx.x == 0) { double* q = part + ((size_t)c * nblk + blockIdx.x) * 3; q[0] = n; q[1] = s1; q[2] = s2; }
}
__global__ void case_stats_final_kernel(const double* __restrict__ part, double* __restrict__ stats, int nblk) {
    const int c = blockIdx.x, j = threadIdx.x;
    if (j >= 3) return;
    double a = 0.0;
    for (int i = 0; i < nblk; ++i) a += part[((size_t)c * nblk + i) * 3 + j];
    stats[c * 3 + j] = a;
}

// batch[k, c, z, y, x] (k-th test-time flip of the padded, normalised crop): source voxel = un-flipped padded coordinate - pad_left inside
// the crop, else the zero padding; EVERY voxel is normalised ((0 - mean) / std in the padding, test.py:113), in float64 then rounded once
__global__ __launch_bounds__(256) void case_prepare_kernel(const float* __restrict__ image, const double* __restrict__ stats, float* __restrict__ batch,
                                                           int C, int D, int H, int W, Box3 b, int pz, int py, int px, int Dp, int Hp, int Wp, int K, unsigned flips) {
    const int kc = blockIdx.y;
    const int c = kc % C, k = kc / C;
    const unsigned f = (flips >> (3 * k)) & 7u;
    const double n = stats[c * 3], mean = stats[c * 3 + 1] / n, mean2 = stats[c * 3 + 2] / n;
    const double sd = sqrt(mean2 - mean * mean);
    const float* p = image + (size_t)c * D * H * W;
    float* dst = batch + (size_t)kc * Dp * Hp * Wp;
    const size_t Vp = (size_t)Dp * Hp * Wp;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < Vp; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % Wp);
        const size_t r = v / Wp;
        const int y = (int)(r % Hp), z = (int)(r / Hp);
        const int zz = ((f & 1u) ? Dp - 1 - z : z) - pz, yy = ((f & 2u) ? Hp - 1 - y : y) - py, xx = ((f & 4u) ? Wp - 1 - x : x) - px;
        float t = 0.f;
        if (zz >= 0 && zz < b.size[0] && yy >= 0 && yy < b.size[1] && xx >= 0 && xx < b.size[2])
            t = p[((size_t)(b.lo[0] + zz) * H + (b.lo[1] + yy)) * W + b.lo[2] + xx];
        dst[v] = (float)(((double)t - mean) / sd);        // a division, as numpy does it (test.py:113)
    }
}

// ------------------------------------------------------------------ TTA merge on a sub-box (test.py:134-144)
// as tta_merge_kernel (pointwise.hip: un-flip, ((p0 + p1) + p2 + ...) / K in float32, > 0.5), for the voxels of `b` only -- the padding
// is dropped here, so the per-class counts are those of the un-padded volume (the ET > 32 rule of test.py:157 counts them)
__global__ __launch_bounds__(256) void tta_merge_box_kernel(const float* __restrict__ p, int K, unsigned flips, float* __restrict__ mean_out,
                                                            unsigned char* __restrict__ mask_out, unsigned long long* __restrict__ counts,
                                                            int C, int D, int H, int W, Box3 b) {
    __shared__ unsigned int cnt[4];
    const size_t Vb = (size_t)b.size[0] * b.size[1] * b.size[2];
    const int c = blockIdx.y;
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned int local = 0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < Vb; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % b.size[2]) + b.lo[2];
        const size_t r = v / b.size[2];
        const int y = (int)(r % b.size[1]) + b.lo[1], z = (int)(r / b.size[1]) + b.lo[0];
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            const unsigned f = (flips >> (3 * k)) & 7u;
            const int zz = (f & 1u) ? D - 1 - z : z, yy = (f & 2u) ? H - 1 - y : y, xx = (f & 4u) ? W - 1 - x : x;
            const float t = p[(((size_t)k * C + c) * D + zz) * H * W + (size_t)yy * W + xx];
            acc = k == 0 ? t : acc + t;
        }
        const float m = acc / (float)K;
        if (mean_out) mean_out[(size_t)c * Vb + v] = m;
        const bool on = m > 0.5f;
        mask_out[(size_t)c * Vb + v] = on ? 1 : 0;
        local += on ? 1u : 0u;
    }
    atomicAdd(&cnt[threadIdx.x >> 6], local);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&counts[c], (unsigned long long)cnt[0] + cnt[1] + cnt[2] + cnt[3]);
}

// ------------------------------------------------------------------ 26-connected components + small-region rejection (test.py:51-62,162-164)
// Label equivalence by union-find on the voxel indices: parent[v] = v for every foreground voxel, each voxel is united with its 13
// "earlier" neighbours of the 26-neighbourhood (the other 13 are covered from the neighbour's side), roots are the smallest index of
// their component (atomicMin), then every voxel is pointed at its root and the roots count their members.  The reference numbers its
// components differently (skimage.morphology.label), but only component SIZES enter test.py:51-62, so the result is identical:
// a foreground voxel survives iff size(component) >= ratio * (V - max(size of any label, background included)).
__device__ __forceinline__ int cc_find(const int* parent, int i) {
    int p = __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // L2-served: other CUs' links are seen
    while (p != i) { i = p; p = __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // parents only ever decrease: terminates
    return i;
}
__device__ __forceinline__ void cc_unite(int* parent, int a, int b) {
    for (;;) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }     // a < b: hang b under a
        const int old = atomicMin(parent + b, a);
        if (old == b) return;                             // b was still a root: linked
        b = old;                                          // somebody re-parented b meanwhile: continue from there (the atomic's value is never stale)
    }
}
// Three passes instead of one (round 5: the plain form -- every voxel united with its 13 earlier neighbours through uncompressed trees -- took 18 ms on the
// 4-million-voxel noise prediction of a random-init network, bench.py's predict_case):
//   init     : a foreground voxel points at the SMALLEST of its earlier foreground neighbours (itself if none): plain reads of the label volume, no
//              atomics -- most of the component's links exist after this pass, as a forest with decreasing indices;
//   compress : every voxel points at the root of its tree (stale parents read on the way are still ancestors: the pass is race-tolerant);
//   merge    : the remaining equivalences -- a voxel and an earlier neighbour whose trees still differ -- by cc_unite on one- or two-hop paths;
//   compress : again, so that counting and the rejection pass find their root in one hop.
__device__ __forceinline__ bool cc_earlier(int dz, int dy, int dx) { return dz < 0 || (dz == 0 && (dy < 0 || (dy == 0 && dx < 0))); }
__global__ __launch_bounds__(256) void cc_init_kernel(const unsigned char* __restrict__ labels, int* __restrict__ parent, int* __restrict__ count, int D, int H, int W) {
    const size_t V = (size_t)D * H * W;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        count[v] = 0;
        if (!labels[v]) { parent[v] = -1; continue; }
        const int x = (int)(v % W);
        const size_t r = v / W;
        const int y = (int)(r % H), z = (int)(r / H);
        int m = (int)v;
#pragma unroll
        for (int dz = -1; dz <= 0; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!cc_earlier(dz, dy, dx)) continue;
                    const int zz = z + dz, yy = y + dy, xx = x + dx;
                    if (zz < 0 || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const size_t u = ((size_t)zz * H + yy) * W + xx;
                    if (labels[u] && (int)u < m) m = (int)u;
                }
        parent[v] = m;
    }
}
__global__ __launch_bounds__(256) void cc_compress_kernel(int* parent, size_t V) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        if (parent[v] < 0) continue;
        const int root = cc_find(parent, (int)v);
        __hip_atomic_store(parent + v, root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // (an ancestor whatever the other threads do meanwhile)
    }
}
__global__ __launch_bounds__(256) void cc_merge_kernel(int* parent, int D, int H, int W) {
    const size_t V = (size_t)D * H * W;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        const int pv = parent[v];
        if (pv < 0) continue;
        const int x = (int)(v % W);
        const size_t r = v / W;
        const int y = (int)(r % H), z = (int)(r / H);
        // the 13 neighbours that precede v in linear order: (dz, dy, dx) with dz = -1, or dz = 0 and dy = -1, or dz = dy = 0 and dx = -1
#pragma unroll
        for (int dz = -1; dz <= 0; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!cc_earlier(dz, dy, dx)) continue;
                    const int zz = z + dz, yy = y + dy, xx = x + dx;
                    if (zz < 0 || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const size_t u = ((size_t)zz * H + yy) * W + xx;
                    const int pu = parent[u];
                    if (pu >= 0 && pu != pv) cc_unite(parent, (int)v, (int)u);       // (equal parents: one tree already; a stale read only costs a redundant unite)
                }
    }
}
// wave-aggregated counting: consecutive voxels mostly share their root, and atomics on one address serialise (~12 ns each: a 100 000-voxel
// tumour counted voxel by voxel would take over a millisecond) -- the lanes of a wave that hold the same root elect one to add their number
__global__ __launch_bounds__(256) void cc_count_kernel(const int* parent, int* count, size_t V) {
    const size_t vend = (V + 255) / 256 * 256;                 // whole waves stay in the loop (the ballots need every lane)
    // ... and a wave carries the (root, number) of its last group across its iterations (wave-uniform), adding it when the root changes: a component
    // that fills the volume -- the noise prediction of a random-init network -- costs one atomic per wave instead of one per 64 voxels (0.72 -> ~0.15 ms)
    int run_root = -1, run_cnt = 0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < vend; v += (size_t)gridDim.x * 256) {
        int root = -1;
        if (v < V && parent[v] >= 0) root = cc_find(parent, (int)v);
        unsigned long long todo = __ballot(root >= 0);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int lroot = __shfl(root, leader);
            const unsigned long long same = __ballot(root == lroot) & todo;
            const int n = (int)__builtin_popcountll(same);
            if (lroot == run_root) run_cnt += n;
            else {
                if (run_cnt && (threadIdx.x & 63) == 0) atomicAdd(count + run_root, run_cnt);
                run_root = lroot; run_cnt = n;
            }
            todo &= ~same;
        }
    }
    if (run_cnt && (threadIdx.x & 63) == 0) atomicAdd(count + run_root, run_cnt);
}
// scal[0] = largest component, scal[1] = number of foreground voxels: one atomic pair per workgroup, at most 256 workgroups
__global__ __launch_bounds__(256) void cc_max_kernel(const int* __restrict__ count, int* __restrict__ scal, size_t V) {
    __shared__ int sm[4][2];
    int mx = 0, fg = 0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        const int c = count[v];
        mx = max(mx, c);
        fg += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = max(mx, __shfl_xor(mx, o)); fg += __shfl_xor(fg, o); }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = mx; sm[threadIdx.x >> 6][1] = fg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = max(max(sm[0][0], sm[1][0]), max(sm[2][0], sm[3][0]));
        fg = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
        if (mx) atomicMax(scal, mx);
        if (fg) atomicAdd(scal + 1, fg);
    }
}
__global__ __launch_bounds__(256) void cc_apply_kernel(unsigned char* __restrict__ labels, const int* __restrict__ parent, const int* __restrict__ count,
                                                       const int* __restrict__ scal, double ratio, size_t V) {
    const long long bg = (long long)V - scal[1];
    const long long biggest = bg > scal[0] ? bg : (long long)scal[0];                // counts.max() over every label, background included (test.py:54-55)
    const double thr = ratio * (double)((long long)V - biggest);                     // c < ratio * nonzero in float64, as numpy evaluates it
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        const int p = parent[v];
        if (p < 0) continue;
        const int root = cc_find(parent, (int)v);
        if ((double)count[root] < thr) labels[v] = 0;
    }
}

// full[D][H][W] = 0 except the box, which takes lab[size] (test.py:167-168)
__global__ __launch_bounds__(256) void paste_labels_kernel(const unsigned char* __restrict__ lab, unsigned char* __restrict__ full, int D, int H, int W, Box3 b) {
    const size_t V = (size_t)D * H * W;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % W) - b.lo[2];
        const size_t r = v / W;
        const int y = (int)(r % H) - b.lo[1], z = (int)(r / H) - b.lo[0];
        unsigned char o = 0;
        if (z >= 0 && z < b.size[0] && y >= 0 && y < b.size[1] && x >= 0 && x < b.size[2]) o = lab[((size_t)z * b.size[1] + y) * b.size[2] + x];
        full[v] = o;
    }
}

static int make_box(Box3& b, const int* lo, const int* size, int D, int H, int W, const char* who) {
    const int dims[3] = {D, H, W};
    for (int a = 0; a < 3; ++a) {
        if (!(lo[a] >= 0 && size[a] > 0 && lo[a] + size[a] <= dims[a])) { set_error("%s: the box must lie inside the volume", who); return RU_EINVAL; }
        b.lo[a] = lo[a]; b.size[a] = size[a];
    }
    return RU_OK;
}

}  // namespace ru

using namespace ru;

extern "C" int ru_tile_gather(const float* data, float* tiles, int N, int C, int D, int H, int W, int T, const int* origins, int td, int th, int tw,
                              ru_stream_t stream) {
    RU_REQUIRE(data && tiles && origins && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && T > 0 && td > 0 && th > 0 && tw > 0, "ru_tile_gather: bad argument");
    RU_REQUIRE((tw & 3) == 0, "ru_tile_gather: the tile width must be a multiple of 4");
    const size_t per_tile = (size_t)N * C * td * th * tw;
    for (int t0 = 0; t0 < T; t0 += kMaxTiles) {
        const int nt = T - t0 < kMaxTiles ? T - t0 : kMaxTiles;
        TileOrigins org;
        for (int t = 0; t < nt; ++t) for (int a = 0; a < 3; ++a) org.o[t][a] = origins[(size_t)(t0 + t) * 3 + a];
        hipLaunchKernelGGL(tile_gather_kernel, dim3(grid1d((size_t)td * th * (tw / 4), 256, 1024), (unsigned)(nt * N * C)), dim3(256), 0, (hipStream_t)stream,
                           data, tiles + (size_t)t0 * per_tile, org, N, C, D, H, W, td, th, tw);
        RU_CHECK_LAUNCH("tile_gather_kernel");
    }
    return RU_OK;
}

extern "C" int ru_tile_scatter(const float* tiles, float* out, int N, int C, int D, int H, int W, int T, const int* origins, int td, int th, int tw,
                               const int* border, const int* center, ru_stream_t stream) {
    RU_REQUIRE(tiles && out && origins && border && center && N > 0 && C > 0 && T > 0, "ru_tile_scatter: bad argument");
    const int ts[3] = {td, th, tw};
    for (int a = 0; a < 3; ++a) RU_REQUIRE(border[a] >= 0 && center[a] > 0 && border[a] + center[a] <= ts[a], "ru_tile_scatter: centre block outside the tile");
    for (int t = 0; t < T; ++t) for (int a = 0; a < 3; ++a) RU_REQUIRE(origins[(size_t)t * 3 + a] + border[a] >= 0, "ru_tile_scatter: centre block before the volume start");
    const size_t per_tile = (size_t)N * C * td * th * tw;
    for (int t0 = 0; t0 < T; t0 += kMaxTiles) {
        const int nt = T - t0 < kMaxTiles ? T - t0 : kMaxTiles;
        TileOrigins org;
        for (int t = 0; t < nt; ++t) for (int a = 0; a < 3; ++a) org.o[t][a] = origins[(size_t)(t0 + t) * 3 + a];
        hipLaunchKernelGGL(tile_scatter_kernel, dim3(grid1d((size_t)center[0] * center[1] * center[2], 256, 1024), (unsigned)(nt * N * C)), dim3(256), 0,
                           (hipStream_t)stream, tiles + (size_t)t0 * per_tile, out, org, N, C, D, H, W, td, th, tw, border[0], border[1], border[2],
                           center[0], center[1], center[2]);
        RU_CHECK_LAUNCH("tile_scatter_kernel");
    }
    return RU_OK;
}

extern "C" int ru_case_bbox(const float* image, int* box, int C, int D, int H, int W, ru_stream_t stream) {
    RU_REQUIRE(image && box && C > 0 && D > 0 && H > 0 && W > 0, "ru_case_bbox: bad argument");
    hipLaunchKernelGGL(bbox_init_kernel, dim3(cdiv(C * 6, 64)), dim3(64), 0, (hipStream_t)stream, box, C);
    RU_CHECK_LAUNCH("bbox_init_kernel");
    hipLaunchKernelGGL(bbox_kernel, dim3(grid1d((size_t)D * H * W, 256 * 8, 128), (unsigned)C), dim3(256), 0, (hipStream_t)stream, image, box, D, H, W);
    RU_CHECK_LAUNCH("bbox_kernel");
    return RU_OK;
}

extern "C" size_t ru_case_workspace_bytes(int C, int D, int H, int W) {
    const size_t V = (size_t)D * H * W;
    return 256 + (size_t)C * ((V + CS_CHUNK - 1) / CS_CHUNK) * 3 * sizeof(double);
}

extern "C" int ru_case_stats(const float* image, double* stats, int C, int D, int H, int W, const int* lo, const int* size, void* ws, size_t ws_bytes,
                             ru_stream_t stream) {
    RU_REQUIRE(image && stats && lo && size && ws && C > 0, "ru_case_stats: bad argument");
    Box3 b;
    int rc = make_box(b, lo, size, D, H, W, "ru_case_stats");
    if (rc) return rc;
    const size_t Vb = (size_t)b.size[0] * b.size[1] * b.size[2];
    const int nblk = (int)((Vb + CS_CHUNK - 1) / CS_CHUNK);
    RU_REQUIRE(ws_bytes >= (size_t)C * nblk * 3 * sizeof(double), "ru_case_stats: workspace too small");
    hipLaunchKernelGGL(case_stats_partial_kernel, dim3(nblk, C), dim3(256), 0, (hipStream_t)stream, image, (double*)ws, D, H, W, b, nblk);
    RU_CHECK_LAUNCH("case_stats_partial_kernel");
    hipLaunchKernelGGL(case_stats_final_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, (const double*)ws, stats, nblk);
    RU_CHECK_LAUNCH("case_stats_final_kernel");
    return RU_OK;
}

extern "C" int ru_case_prepare(const float* image, const double* stats, float* batch, int C, int D, int H, int W, const int* lo, const int* size,
                               const int* pad_left, const int* padded, int K, unsigned flips, ru_stream_t stream) {
    RU_REQUIRE(image && stats && batch && lo && size && pad_left && padded && C > 0 && K >= 1 && K <= 8, "ru_case_prepare: bad argument");
    Box3 b;
    int rc = make_box(b, lo, size, D, H, W, "ru_case_prepare");
    if (rc) return rc;
    for (int a = 0; a < 3; ++a) RU_REQUIRE(pad_left[a] >= 0 && pad_left[a] + size[a] <= padded[a], "ru_case_prepare: the crop does not fit the padded extent");
    hipLaunchKernelGGL(case_prepare_kernel, dim3(grid1d((size_t)padded[0] * padded[1] * padded[2], 256 * 4, 4096), (unsigned)(K * C)), dim3(256), 0,
                       (hipStream_t)stream, image, stats, batch, C, D, H, W, b, pad_left[0], pad_left[1], pad_left[2], padded[0], padded[1], padded[2], K, flips);
    RU_CHECK_LAUNCH("case_prepare_kernel");
    return RU_OK;
}

extern "C" int ru_tta_merge_box(const float* probs, int K, unsigned flips, float* mean_out, unsigned char* mask, unsigned long long* counts,
                                int C, int D, int H, int W, const int* lo, const int* size, ru_stream_t stream) {
    RU_REQUIRE(probs && mask && counts && lo && size && K >= 1 && K <= 8 && C >= 1, "ru_tta_merge_box: bad argument");
    Box3 b;
    int rc = make_box(b, lo, size, D, H, W, "ru_tta_merge_box");
    if (rc) return rc;
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(unsigned long long) * C, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(counts)");
    hipLaunchKernelGGL(tta_merge_box_kernel, dim3(grid1d((size_t)size[0] * size[1] * size[2], 256 * 8, 512), C), dim3(256), 0, (hipStream_t)stream,
                       probs, K, flips, mean_out, mask, counts, C, D, H, W, b);
    RU_CHECK_LAUNCH("tta_merge_box_kernel");
    return RU_OK;
}

extern "C" size_t ru_cc_workspace_bytes(int D, int H, int W) { return 512 + 2 * align_up((size_t)D * H * W * sizeof(int), 256); }

extern "C" int ru_cc_reject(unsigned char* labels, int D, int H, int W, double ratio, void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(labels && ws && D > 0 && H > 0 && W > 0 && ratio >= 0.0, "ru_cc_reject: bad argument");
    const size_t V = (size_t)D * H * W;
    RU_REQUIRE(V < (size_t)INT_MAX, "ru_cc_reject: volume too large for 32-bit voxel indices");
    RU_REQUIRE(ws_bytes >= ru_cc_workspace_bytes(D, H, W), "ru_cc_reject: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int* scal = (int*)ws;
    int* parent = (int*)((char*)ws + 256);
    int* count = (int*)((char*)ws + 256 + align_up(V * sizeof(int), 256));
    hipError_t e = hipMemsetAsync(scal, 0, 256, s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(cc)");
    const unsigned g = grid1d(V, 256 * 4, 4096);
    hipLaunchKernelGGL(cc_init_kernel, dim3(g), dim3(256), 0, s, labels, parent, count, D, H, W);
    RU_CHECK_LAUNCH("cc_init_kernel");
    hipLaunchKernelGGL(cc_compress_kernel, dim3(g), dim3(256), 0, s, parent, V);
    RU_CHECK_LAUNCH("cc_compress_kernel");
    hipLaunchKernelGGL(cc_merge_kernel, dim3(g), dim3(256), 0, s, parent, D, H, W);
    RU_CHECK_LAUNCH("cc_merge_kernel");
    hipLaunchKernelGGL(cc_compress_kernel, dim3(g), dim3(256), 0, s, parent, V);
    RU_CHECK_LAUNCH("cc_compress_kernel");
    hipLaunchKernelGGL(cc_count_kernel, dim3(g), dim3(256), 0, s, parent, count, V);
    RU_CHECK_LAUNCH("cc_count_kernel");
    hipLaunchKernelGGL(cc_max_kernel, dim3(g < 256 ? g : 256), dim3(256), 0, s, count, scal, V);
    RU_CHECK_LAUNCH("cc_max_kernel");
    hipLaunchKernelGGL(cc_apply_kernel, dim3(g), dim3(256), 0, s, labels, parent, count, scal, ratio, V);
    RU_CHECK_LAUNCH("cc_apply_kernel");
    return RU_OK;
}

extern "C" int ru_paste_labels(const unsigned char* lab, unsigned char* full, int D, int H, int W, const int* lo, const int* size, ru_stream_t stream) {
    RU_REQUIRE(lab && full && lo && size, "ru_paste_labels: null argument");
    Box3 b;
    int rc = make_box(b, lo, size, D, H, W, "ru_paste_labels");
    if (rc) return rc;
    hipLaunchKernelGGL(paste_labels_kernel, dim3(grid1d((size_t)D * H * W, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream, lab, full, D, H, W, b);
    RU_CHECK_LAUNCH("paste_labels_kernel");
    return RU_OK;
}
