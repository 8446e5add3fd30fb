// Microtest (GPU box): is the 6-term bf16x3 split product on v_mfma_f32_32x32x16_bf16 bitwise reproducible?
// Every wave of every workgroup computes the same 32x32 tile from the same operands; the host compares all copies.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/split_mfma_test.hip -o /tmp/split_test && /tmp/split_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Bf16x3 { bf16x8_t p1, p2, p3; };
__device__ __forceinline__ void split2(float a, float b, bf16x2_t& q1, bf16x2_t& q2, bf16x2_t& q3) {
    const f32x2_t v = {a, b};
    q1 = __builtin_convertvector(v, bf16x2_t);
    const f32x2_t r1 = v - __builtin_convertvector(q1, f32x2_t);
    q2 = __builtin_convertvector(r1, bf16x2_t);
    const f32x2_t r2 = r1 - __builtin_convertvector(q2, f32x2_t);
    q3 = __builtin_convertvector(r2, bf16x2_t);
}
__device__ __forceinline__ Bf16x3 split8(const float4& lo, const float4& hi) {
    bf16x2_t a1, a2, a3, b1, b2, b3, c1, c2, c3, d1, d2, d3;
    split2(lo.x, lo.y, a1, a2, a3); split2(lo.z, lo.w, b1, b2, b3);
    split2(hi.x, hi.y, c1, c2, c3); split2(hi.z, hi.w, d1, d2, d3);
    Bf16x3 o;
    o.p1 = bf16x8_t{a1.x, a1.y, b1.x, b1.y, c1.x, c1.y, d1.x, d1.y};
    o.p2 = bf16x8_t{a2.x, a2.y, b2.x, b2.y, c2.x, c2.y, d2.x, d2.y};
    o.p3 = bf16x8_t{a3.x, a3.y, b3.x, b3.y, c3.x, c3.y, d3.x, d3.y};
    return o;
}
template <int MODE>
__global__ __launch_bounds__(512) void k(const float4* w, const float4* x, float* out, int npairs) {
    const int lane = threadIdx.x & 63;
    f32x16 acc, acc_s;
    for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc_s[q] = 0.f; }
    for (int p = 0; p < npairs; ++p) {
        const float4 w0 = w[(2 * p) * 64 + lane], w1 = w[(2 * p + 1) * 64 + lane];
        const float4 x0 = x[(2 * p) * 64 + lane], x1 = x[(2 * p + 1) * 64 + lane];
        const Bf16x3 w_ = split8(w0, w1), x_ = split8(x0, x1);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_.p1, x_.p1, acc, 0, 0, 0);
        if (MODE >= 1) {
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_.p1, x_.p2, acc_s, 0, 0, 0);
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_.p2, x_.p1, acc_s, 0, 0, 0);
        }
        if (MODE >= 2) {
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_.p2, x_.p2, acc_s, 0, 0, 0);
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_.p1, x_.p3, acc_s, 0, 0, 0);
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_.p3, x_.p1, acc_s, 0, 0, 0);
        }
    }
    float* o = out + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 1024;
    for (int q = 0; q < 16; ++q) o[q * 64 + lane] = acc[q] + acc_s[q];
}
template <int MODE>
int run(const float4* dw, const float4* dx, float* dout, int npairs, const char* tag) {
    const int G = 512, copies = G * 8;
    std::vector<float> h((size_t)copies * 1024), ref(1024);
    int bad_total = 0;
    for (int rep = 0; rep < 5; ++rep) {
        k<MODE><<<G, 512>>>(dw, dx, dout, npairs);
        (void)hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
        if (rep == 0) std::memcpy(ref.data(), h.data(), 4096);
        int bad = 0;
        for (int c = 0; c < copies; ++c) bad += std::memcmp(ref.data(), h.data() + (size_t)c * 1024, 4096) != 0;
        bad_total += bad;
        printf("%s rep %d: %d of %d copies differ from copy 0 of rep 0\n", tag, rep, bad, copies);
    }
    return bad_total;
}
int main() {
    const int npairs = 14;
    std::vector<float> hw(2 * npairs * 64 * 4), hx(hw.size());
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hw) v = rnd() * 0.1f;
    for (auto& v : hx) v = rnd() * 2.f;
    float4 *dw, *dx; float* dout;
    (void)hipMalloc(&dw, hw.size() * 4); (void)hipMalloc(&dx, hx.size() * 4); (void)hipMalloc(&dout, (size_t)512 * 8 * 1024 * 4);
    (void)hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    int bad = run<0>(dw, dx, dout, npairs, "1 term ");
    bad += run<1>(dw, dx, dout, npairs, "3 terms");
    bad += run<2>(dw, dx, dout, npairs, "6 terms");
    printf(bad ? "NOT reproducible\n" : "reproducible\n");
    return bad != 0;
}
