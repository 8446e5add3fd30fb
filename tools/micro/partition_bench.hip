// Microbenchmark (GPU box): how long does a launch shaped like the decoder LSTM step take when its bytes are dealt to the
// workgroups in different ways - 256 tiles as today (128 x 224 KB + 128 x 320 KB), the same 256 tiles while another kernel
// holds 32 CUs, and 224 workgroups with a balanced deal (96 x 320, 64 x 272, 64 x 336 KB) beside that kernel.  Every
// workgroup streams its own slice of a 70-MB buffer (1 KiB per wave-load, 4 in flight per wave, 4 fp32 MFMAs per load) and,
// like the real kernel, one x fragment per weight fragment from a small shared buffer that stays in L2.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/partition_bench.hip -o /tmp/partition_bench && /tmp/partition_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int WAVES = 8, DEPTH = 4;

struct Deal { int first_group; int groups; int x_num, x_den; };   // per workgroup: slice of the weight buffer (KiB groups), x loads per W load

__global__ __launch_bounds__(WAVES * 64) void tile_kernel(const float4* w, const float4* x, const Deal* deals, float* out, int x_groups) {
    const Deal d = deals[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per = (d.groups + WAVES - 1) / WAVES;
    const int g0 = min(d.groups, wave * per), g1 = min(d.groups, g0 + per);
    const float4* p = w + (long)(d.first_group + g0) * 64 + lane;
    const int n = g1 - g0;
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    if (n > 0) {
        float4 v[DEPTH], xv[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            v[u] = p[(long)min(u, n - 1) * 64];
            xv[u] = x[(long)(((g0 + u) * d.x_num / d.x_den) % x_groups) * 64 + lane];
        }
        for (int base = 0; base < n; base += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].x, xv[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].y, xv[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].z, xv[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].w, xv[u].w, acc, 0, 0, 0);
                const int g = min(base + u + DEPTH, n - 1);
                v[u] = p[(long)g * 64];
                // x_num / x_den < 1: consecutive W groups share an x fragment (several column tiles per pass over x): the
                // repeated address is served by L1, which is what holding the fragment in registers would amount to
                xv[u] = x[(long)(((g0 + g) * d.x_num / d.x_den) % x_groups) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __shared__ float red[WAVES * 16 * 64];
    for (int q = 0; q < 16; ++q) red[(wave * 16 + q) * 64 + lane] = acc[q];
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < WAVES * 16 * 64; i += WAVES * 64) s += red[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// holds `gridDim.x` CUs (a whole CU each: 1024 threads, 160 KB of LDS) until *flag != 0 or ~limit_us have passed
__global__ __launch_bounds__(1024) void blocker_kernel(const unsigned* flag, unsigned* arrived, unsigned limit_us) {
    extern __shared__ float pad[];
    if (threadIdx.x == 0) {
        pad[0] = 1.f;
        __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u &&
               (wall_clock64() - t0) < (unsigned long long)limit_us * 100ull)
            __builtin_amdgcn_s_sleep(16);
    }
    __syncthreads();
}

static float time_launches(int grid, const float4* w, const float4* x, const Deal* deals, float* out, int x_groups, hipStream_t s) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) tile_kernel<<<grid, WAVES * 64, 0, s>>>(w, x, deals, out, x_groups);
    (void)hipEventRecord(a, s);
    const int reps = 40;
    for (int i = 0; i < reps; ++i) tile_kernel<<<grid, WAVES * 64, 0, s>>>(w, x, deals, out, x_groups);
    (void)hipEventRecord(b, s);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    const long total_groups = 70L * 1024;   // KiB
    const int x_groups = 320;
    float4 *w, *x; float* out; Deal* dd; unsigned* flags;
    (void)hipMalloc(&w, total_groups * 1024);
    (void)hipMalloc(&x, x_groups * 1024);
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&dd, 512 * sizeof(Deal));
    (void)hipMalloc(&flags, 256);
    (void)hipMemset(w, 0, total_groups * 1024);
    (void)hipMemset(x, 0, x_groups * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blocker_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t sa, sb;
    (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);

    auto deal = [&](const std::vector<Deal>& sizes) {
        std::vector<Deal> v = sizes;
        int at = 0;
        for (auto& d : v) { d.first_group = at; at += d.groups; }
        (void)hipMemcpy(dd, v.data(), v.size() * sizeof(Deal), hipMemcpyHostToDevice);
        return at;
    };
    auto with_blocker = [&](int n_block, auto&& body) {
        (void)hipMemset(flags, 0, 256);
        if (n_block) {
            blocker_kernel<<<n_block, 1024, 160 * 1024, sb>>>(flags, flags + 32, 200000u);
            unsigned arrived = 0;
            for (int i = 0; i < 1000 && arrived < (unsigned)n_block; ++i) (void)hipMemcpy(&arrived, flags + 32, 4, hipMemcpyDeviceToHost);
            printf("  (%u of %d blocking workgroups resident)\n", arrived, n_block);
        }
        body();
        const unsigned one = 1;
        (void)hipMemcpy(flags, &one, 4, hipMemcpyHostToDevice);
        (void)hipStreamSynchronize(sb);
    };

    std::vector<Deal> today, balanced, balanced_1to1;
    for (int i = 0; i < 128; ++i) today.push_back({0, 224, 1, 1});
    for (int i = 0; i < 128; ++i) today.push_back({0, 320, 1, 1});
    for (int i = 0; i < 96; ++i) { balanced.push_back({0, 320, 1, 1}); balanced_1to1.push_back({0, 320, 1, 1}); }
    for (int i = 0; i < 64; ++i) { balanced.push_back({0, 272, 1, 1}); balanced_1to1.push_back({0, 272, 1, 1}); }
    for (int i = 0; i < 64; ++i) { balanced.push_back({0, 336, 2, 3}); balanced_1to1.push_back({0, 336, 1, 1}); }   // 3 half tiles share x: 2 x loads per 3 W loads

    int kb = deal(today);
    with_blocker(0, [&] { printf("256 tiles (128 x 224 + 128 x 320 KB), whole chip:              %6.2f us/launch (%d KB)\n", time_launches(256, w, x, dd, out, x_groups, sa), kb); });
    with_blocker(32, [&] { printf("256 tiles, 32 CUs held by another kernel:                    %6.2f us/launch\n", time_launches(256, w, x, dd, out, x_groups, sa)); });
    kb = deal(balanced);
    with_blocker(32, [&] { printf("224 workgroups 96 x 320 / 64 x 272 / 64 x 336 KB, 32 CUs held: %6.2f us/launch (%d KB)\n", time_launches(224, w, x, dd, out, x_groups, sa), kb); });
    with_blocker(0, [&] { printf("224 workgroups, same deal, whole chip:                       %6.2f us/launch\n", time_launches(224, w, x, dd, out, x_groups, sa)); });
    kb = deal(balanced_1to1);
    with_blocker(32, [&] { printf("224 workgroups, same deal with one x load per W load, 32 held: %6.2f us/launch\n", time_launches(224, w, x, dd, out, x_groups, sa)); });
    return 0;
}
