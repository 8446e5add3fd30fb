// Microbenchmark (GPU box): ceiling of the decoder LSTM step's weight stream.  256 workgroups each stream their own
// contiguous slice of a 71-MB buffer that is re-read every launch (Infinity-Cache resident, like the recurrent weights),
// 1 KiB per wave-load, DEPTH loads in flight per wave, optionally with 4 fp32 MFMAs per load like the real kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/stream_bench.hip -o /tmp/stream_bench && /tmp/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int WAVES, int DEPTH, bool MFMA>
__global__ __launch_bounds__(WAVES * 64) void stream_kernel(const float4* w, float* out, int groups_per_wave) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float4* p = w + ((long)blockIdx.x * WAVES + wave) * groups_per_wave * 64 + lane;
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float4 v[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) v[u] = p[(long)u * 64];
    int base = 0;
    for (; base + 2 * DEPTH <= groups_per_wave; base += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (MFMA) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].x, v[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].y, v[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].z, v[u].w, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].w, v[u].x, acc, 0, 0, 0);
            } else {
                acc[u & 15] += v[u].x + v[u].y + v[u].z + v[u].w;
            }
            v[u] = p[(long)(base + u + DEPTH) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) acc[u & 15] += v[u].x + v[u].w;
    float s = 0.f;
    for (int q = 0; q < 16; ++q) s += acc[q];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int WAVES, int DEPTH, bool MFMA>
void run(const float4* w, float* out, long total_groups, const char* tag) {
    const int gpw = (int)(total_groups / (256L * WAVES));  // groups (KiB) per wave
    hipEvent_t a, b;
    (void)(void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) stream_kernel<WAVES, DEPTH, MFMA><<<256, WAVES * 64>>>(w, out, gpw);
    (void)hipEventRecord(a);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) stream_kernel<WAVES, DEPTH, MFMA><<<256, WAVES * 64>>>(w, out, gpw);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = 256.0 * WAVES * gpw * 1024.0;
    printf("%-28s waves %2d depth %2d  %7.2f us/launch  %6.2f TB/s  (%.1f MB)\n", tag, WAVES, DEPTH, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12, bytes / 1e6);
}

int main() {
    const long total_groups = 69632;  // 1-KiB fragments = 71.3 MB, the two decoder LSTM matrices
    float4* w; float* out;
    (void)hipMalloc(&w, total_groups * 1024 + (1 << 20));
    (void)hipMalloc(&out, 4096);
    (void)hipMemset(w, 0, total_groups * 1024 + (1 << 20));
    run<8, 4, false>(w, out, total_groups, "load only");
    run<8, 8, false>(w, out, total_groups, "load only");
    run<8, 12, false>(w, out, total_groups, "load only");
    run<8, 16, false>(w, out, total_groups, "load only");
    run<16, 4, false>(w, out, total_groups, "load only");
    run<16, 8, false>(w, out, total_groups, "load only");
    run<8, 8, true>(w, out, total_groups, "load + 4 MFMA/KiB");
    run<8, 12, true>(w, out, total_groups, "load + 4 MFMA/KiB");
    run<8, 16, true>(w, out, total_groups, "load + 4 MFMA/KiB");
    run<16, 8, true>(w, out, total_groups, "load + 4 MFMA/KiB");
    return 0;
}
