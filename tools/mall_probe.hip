// Probe: what does the 256 MiB Infinity Cache deliver to streaming kernels?  (a) read BW of a buffer of S MiB read repeatedly,
// (b) write S MiB then read it back, for S = 32 .. 1024.  Build: hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ p, size_t n, float* out) {
    float4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = p[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void wr(float4* __restrict__ p, size_t n, float v) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = float4{v, v, v, v};
}
__global__ __launch_bounds__(256) void cp(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = a[i]; v.x += 1.f; b[i] = v; }
}
int main() {
    const size_t MAXB = 2048ull << 20;
    float4 *a, *b; float* o;
    hipMalloc(&a, MAXB); hipMalloc(&b, MAXB); hipMalloc(&o, 64);
    hipMemset(a, 0, MAXB); hipMemset(b, 0, MAXB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int G = 256 * 8;
    for (size_t mb : {16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048}) {
        size_t n = (mb << 20) / 16;
        float ms;
        // repeated read
        for (int i = 0; i < 3; ++i) rd<<<G, 256>>>(a, n, o);
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) rd<<<G, 256>>>(a, n, o); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); double rr = 10.0 * (mb << 20) / (ms * 1e-3) / 1e12;
        // write then read (pairs)
        for (int i = 0; i < 2; ++i) { wr<<<G, 256>>>(a, n, 1.f); rd<<<G, 256>>>(a, n, o); }
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) { wr<<<G, 256>>>(a, n, 1.f); rd<<<G, 256>>>(a, n, o); } hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); double wrr = 10.0 * 2 * (mb << 20) / (ms * 1e-3) / 1e12;
        // write only
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) wr<<<G, 256>>>(a, n, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); double ww = 10.0 * (mb << 20) / (ms * 1e-3) / 1e12;
        // copy a -> b then b -> a (chain: the consumer reads what the producer just wrote)
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) { cp<<<G, 256>>>(a, b, n); cp<<<G, 256>>>(b, a, n); } hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); double cc = 10.0 * 2 * (mb << 20) / (ms * 1e-3) / 1e12;
        printf("%5zu MiB: repeated read %.2f TB/s | write+read-back %.2f TB/s | write only %.2f TB/s | ping-pong copy %.2f TB/s (r+w)\n", mb, rr, wrr, ww, cc);
    }
    return 0;
}
