// Probe: what does a dependent chain of short kernels cost per step when every step also has kernels on a SECOND stream tied in
// with events (fork after one chain kernel, join before a later one) - eager and as a replayed hipGraph?
//     hipcc --offload-arch=gfx950 -O2 chain_sidestream_bench.hip -o chain_sidestream_bench && ./chain_sidestream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void busy(int ns, unsigned* sink) {   // every workgroup spins for ~ns
    const unsigned long long t0 = wall_clock64();   // 100 MHz
    while ((wall_clock64() - t0) * 10ull < (unsigned long long)ns) __builtin_amdgcn_s_sleep(2);
    if (ns < 0) sink[0] = 1;
}

static int enqueue(hipStream_t s, hipStream_t side, hipEvent_t* ev, int steps, bool with_side, unsigned* sink) {
    for (int t = 0; t < steps; ++t) {
        hipEvent_t* e = ev + 6 * (t & 1);   // events are re-recorded every other step
        busy<<<128, 512, 0, s>>>(5800, sink);                          // A'
        if (with_side) { CK(hipEventRecord(e[0], s)); CK(hipStreamWaitEvent(side, e[0], 0)); busy<<<128, 512, 0, side>>>(8000, sink); CK(hipEventRecord(e[1], side)); }   // P1a
        busy<<<16, 512, 0, s>>>(7000, sink);                           // attention
        if (with_side) { CK(hipEventRecord(e[2], s)); CK(hipStreamWaitEvent(side, e[2], 0)); busy<<<192, 512, 0, side>>>(9500, sink); CK(hipEventRecord(e[3], side)); }   // P1b + P3
        if (with_side) CK(hipStreamWaitEvent(s, e[1], 0));
        busy<<<128, 512, 0, s>>>(6600, sink);                          // C'
        if (with_side) { CK(hipEventRecord(e[4], s)); CK(hipStreamWaitEvent(side, e[4], 0)); busy<<<128, 512, 0, side>>>(8000, sink); CK(hipEventRecord(e[5], side)); }   // P2
        busy<<<4, 1024, 0, s>>>(4900, sink);                           // D
        if (with_side) { CK(hipStreamWaitEvent(s, e[3], 0)); CK(hipStreamWaitEvent(s, e[5], 0)); }
    }
    return 0;
}

int main() {
    hipStream_t s, side;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ev[12];
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    const int steps = 64;
    for (int with_side = 0; with_side < 2; ++with_side) {
        for (int rep = 0; rep < 2; ++rep) {   // eager
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            if (enqueue(s, side, ev, steps, with_side, sink)) return 1;
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("%s, eager:  %.2f us per step (kernel time on the chain 24.3 us)\n", with_side ? "chain + side stream" : "chain only", us / steps);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        if (enqueue(s, side, ev, steps, with_side, sink)) return 1;
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, s));
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("%s, graph:  %.2f us per step\n", with_side ? "chain + side stream" : "chain only", us / steps);
        }
    }
    return 0;
}
