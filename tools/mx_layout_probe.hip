// Probe: semantics the MX product scheme of conv3_mx_kernel relies on, checked on the hardware against a host model.
//   (1) v_cvt_pk_fp8_f32 / v_cvt_scalef32_pk_fp8_f32: OCP e4m3, round to nearest even, SATURATING (no NaN / inf for finite inputs)?  What does the
//       scale operand of the second do (multiply or divide)?
//   (2) v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands: lane (row = l & 15, k-group g = l >> 4) holds K elements [32 g, 32 g + 32) as the 32 bytes of its
//       8 registers in order?  (Packing A and B by the same (lane, byte) -> k rule makes the product independent of the true order; what is checked is
//       row = l & 15 for both operands and the standard 16x16 C/D layout.)  Scale operands: E8M0 bytes, 127 = 2^0; the product is scaled by 2^(sa-127) * 2^(sb-127).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mx_layout_probe.hip -o /tmp/mx_layout && /tmp/mx_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__global__ void cvt_kernel(const float* x, int n, unsigned* plain, unsigned* scaled, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    plain[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false) & 0xffffu;
    const s16x2 w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(s16x2{0, 0}, x[2 * i], x[2 * i + 1], scale, false);
    scaled[i] = (unsigned)(unsigned short)w[0];
}

// a: [16][128] bytes (row-major, e4m3), b: [128][16] stored as bt[16][128]; out: [16][16] floats, D[m][n]
// ea / eb: E8M0 exponent bytes PER K-GROUP (packed: byte g of the int) -- a lane passes the scale of its own 32 K elements, replicated in the four bytes of
// its scale register (so the op_sel byte selection does not matter)
__global__ void mfma_kernel(const uint8_t* a, const uint8_t* bt, float* out, unsigned ea4, unsigned eb4) {
    const int lane = threadIdx.x, row = lane & 15, g = lane >> 4;
    const int sa = (int)(((ea4 >> (8 * g)) & 0xffu) * 0x01010101u), sb = (int)(((eb4 >> (8 * g)) & 0xffu) * 0x01010101u);
    i32x8 av, bv;
    for (int j = 0; j < 8; ++j) {
        av[j] = *reinterpret_cast<const int*>(a + row * 128 + 32 * g + 4 * j);
        bv[j] = *reinterpret_cast<const int*>(bt + row * 128 + 32 * g + 4 * j);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + row] = c[r];      // standard 16x16 C/D: col = lane & 15, row = 4 (lane >> 4) + r
}

static float e4m3_decode(uint8_t b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf(1.f + m / 8.f, e - 7);
    return s ? -v : v;
}
static uint8_t e4m3_encode_sat(float x) {      // nearest representable (ties to even mantissa), saturating at 448
    if (std::isnan(x)) return 0x7f;
    const uint8_t s = std::signbit(x) ? 0x80 : 0;
    float a = fabsf(x);
    if (a >= 448.f) return s | 0x7e;
    int best = 0;
    float bd = 1e30f;
    for (int c = 0; c < 0x7f; ++c) {
        const float d = fabsf(e4m3_decode((uint8_t)c) - a);
        if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = c; }
    }
    return s | (uint8_t)best;
}

int main() {
    // ---- (1) conversions
    std::vector<float> xs = {0.f, 1.f, -1.f, 0.3f, 1.7f, 1.0625f, 1.1875f, 447.f, 448.f, 449.f, 464.f, 480.f, 500.f, 1e4f, -1e6f, 1e-3f, 0.001953125f, 0.0009765625f,
                             0.00146484375f, 0.015625f, 0.0146f, 3.3e-5f, -0.0123f, 17.f, 18.f, 19.f, 20.f, 21.f, 22.f, 23.f, 27.f, 29.f, 208.f, 216.f, 224.f, 232.f, 240.f, 7.3e-4f};
    uint32_t st = 777u;
    while (xs.size() < 4096) { st = st * 1664525u + 1013904223u; const float u = (float)(st >> 8) / 16777216.f; xs.push_back(ldexpf(u * 2.f - 1.f, (int)(st & 15) - 8)); }
    const int n = (int)xs.size();
    float* dx; unsigned *dp, *ds;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dp, n * 2); (void)hipMalloc(&ds, n * 2);
    (void)hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
    const float scale = 0.25f;
    cvt_kernel<<<(n / 2 + 255) / 256, 256>>>(dx, n, dp, ds, scale);
    std::vector<unsigned> hp(n / 2), hs(n / 2);
    (void)hipMemcpy(hp.data(), dp, n * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(hs.data(), ds, n * 2, hipMemcpyDeviceToHost);
    int bad_plain = 0, bad_div = 0, bad_mul = 0;
    for (int i = 0; i < n; ++i) {
        const uint8_t got = (hp[i / 2] >> (8 * (i & 1))) & 0xff, want = e4m3_encode_sat(xs[i]);
        const uint8_t gs = (hs[i / 2] >> (8 * (i & 1))) & 0xff;
        if (got != want) { if (bad_plain < 12) printf("  cvt_pk_fp8_f32(%g) = 0x%02x (%g), host model 0x%02x (%g)\n", xs[i], got, e4m3_decode(got), want, e4m3_decode(want)); ++bad_plain; }
        if (gs != e4m3_encode_sat(xs[i] / scale)) ++bad_div;
        if (gs != e4m3_encode_sat(xs[i] * scale)) ++bad_mul;
    }
    printf("(1) v_cvt_pk_fp8_f32 against the RNE + saturating OCP e4m3 model: %d of %d differ\n", bad_plain, n);
    printf("    v_cvt_scalef32_pk_fp8_f32(scale = %g): model x / scale differs in %d, model x * scale differs in %d of %d\n", scale, bad_div, bad_mul, n);

    // ---- (2) the scaled MFMA
    std::vector<uint8_t> a(16 * 128), bt(16 * 128);
    for (auto& v : a) { st = st * 1664525u + 1013904223u; v = (uint8_t)((st >> 9) & 0xff); if ((v & 0x7f) == 0x7f) v ^= 1; if ((v & 0x78) > 0x50) v &= 0xbf; }
    for (auto& v : bt) { st = st * 1664525u + 1013904223u; v = (uint8_t)((st >> 9) & 0xff); if ((v & 0x7f) == 0x7f) v ^= 1; if ((v & 0x78) > 0x50) v &= 0xbf; }
    uint8_t *da, *db; float* dout;
    (void)hipMalloc(&da, a.size()); (void)hipMalloc(&db, bt.size()); (void)hipMalloc(&dout, 256 * 4);
    (void)hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice); (void)hipMemcpy(db, bt.data(), bt.size(), hipMemcpyHostToDevice);
    for (int trial = 0; trial < 4; ++trial) {
        // trial 3: scales that differ per k-group (what conv3_mx_kernel uses: k-groups 0 / 2 carry one cross term, 1 / 3 the other)
        const unsigned ea4 = trial == 0 ? 0x7f7f7f7fu : (trial == 1 ? 0x78787878u : (trial == 2 ? 0x7f7f7f7fu : 0x6d776d77u));
        const unsigned eb4 = trial == 2 ? 0x73737373u : (trial == 3 ? 0x7f737f73u : 0x7f7f7f7fu);
        mfma_kernel<<<1, 64>>>(da, db, dout, ea4, eb4);
        std::vector<float> out(256);
        (void)hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
        double worst = 0, mag = 0;
        for (int m = 0; m < 16; ++m)
            for (int nn = 0; nn < 16; ++nn) {
                double r = 0;
                for (int k = 0; k < 128; ++k)
                    r += (double)e4m3_decode(a[m * 128 + k]) * (double)e4m3_decode(bt[nn * 128 + k]) * ldexp(1.0, (int)((ea4 >> (8 * (k / 32))) & 0xff) - 127) * ldexp(1.0, (int)((eb4 >> (8 * (k / 32))) & 0xff) - 127);
                worst = fmax(worst, fabs(r - out[m * 16 + nn]));
                mag = fmax(mag, fabs(r));
            }
        if (trial == 3) {
            for (int m = 0; m < 2; ++m)
                for (int nn = 0; nn < 2; ++nn) {
                    double pg[4] = {0, 0, 0, 0};
                    for (int k = 0; k < 128; ++k) pg[k / 32] += (double)e4m3_decode(a[m * 128 + k]) * (double)e4m3_decode(bt[nn * 128 + k]);
                    printf("    D[%d][%d] = %.9g; raw block sums %.6g %.6g %.6g %.6g\n", m, nn, out[m * 16 + nn], pg[0], pg[1], pg[2], pg[3]);
                }
        }
        printf("(2) mfma_scale 16x16x128 e4m3 x e4m3, scale bytes per k-group A %08x B %08x: max |D - reference| = %.3g (max |reference| %.3g)\n", ea4, eb4, worst, mag);
    }
    return 0;
}
