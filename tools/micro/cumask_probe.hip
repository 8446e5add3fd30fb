// Probe: does hipExtStreamCreateWithCUMask confine a stream's workgroups to the masked CUs on this box, and do two masked
// streams run their kernels at the same time?   hipcc --offload-arch=gfx950 -O2 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void where(unsigned* out, unsigned long long* t, int spin) {
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
        t[2 * blockIdx.x] = wall_clock64();
        for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
        t[2 * blockIdx.x + 1] = wall_clock64();
    }
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d\n", p.multiProcessorCount);
    const int words = (p.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> ma(words, 0), mb(words, 0);
    // stream A: the first quarter of the CU bits, stream B: the rest
    for (int i = 0; i < p.multiProcessorCount; ++i) (i < p.multiProcessorCount / 4 ? ma : mb)[i / 32] |= 1u << (i % 32);
    hipStream_t sa, sb;
    CK(hipExtStreamCreateWithCUMask(&sa, words, ma.data()));
    CK(hipExtStreamCreateWithCUMask(&sb, words, mb.data()));
    const int n = 512;
    unsigned *oa, *ob; unsigned long long *ta, *tb;
    CK(hipMalloc(&oa, n * 4)); CK(hipMalloc(&ob, n * 4)); CK(hipMalloc(&ta, n * 16)); CK(hipMalloc(&tb, n * 16));
    for (int rep = 0; rep < 2; ++rep) {
        where<<<n, 64, 0, sa>>>(oa, ta, 200);
        where<<<n, 64, 0, sb>>>(ob, tb, 200);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned> ha(n), hb(n); std::vector<unsigned long long> hta(2 * n), htb(2 * n);
    CK(hipMemcpy(ha.data(), oa, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), ob, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hta.data(), ta, n * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(htb.data(), tb, n * 16, hipMemcpyDeviceToHost));
    auto cu_of = [](unsigned v) { return (v >> 16) << 12 | ((v >> 8) & 0xf) | ((v >> 12) & 0x1) << 4 | ((v >> 13) & 0x7) << 5; };   // xcc, cu id, sh, se
    std::set<unsigned> sa_cus, sb_cus;
    for (auto v : ha) sa_cus.insert(cu_of(v));
    for (auto v : hb) sb_cus.insert(cu_of(v));
    int common = 0; for (auto c : sa_cus) common += sb_cus.count(c);
    unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
    for (int i = 0; i < n; ++i) { a0 = std::min(a0, hta[2 * i]); a1 = std::max(a1, hta[2 * i + 1]); b0 = std::min(b0, htb[2 * i]); b1 = std::max(b1, htb[2 * i + 1]); }
    printf("stream A ran on %zu distinct CUs, stream B on %zu, in common %d\n", sa_cus.size(), sb_cus.size(), common);
    printf("A: %llu .. %llu   B: %llu .. %llu  (100 MHz ticks; overlapping = concurrent)\n", a0 - std::min(a0, b0), a1 - std::min(a0, b0), b0 - std::min(a0, b0), b1 - std::min(a0, b0));
    return 0;
}
