// Probe: does a kernel of plain fp32 VALU + LDS work stay bit-reproducible while ANOTHER kernel full of
// v_mfma_f32_32x32x16_bf16 (or, for comparison, v_mfma_f32_32x32x2_f32 / no matrix instructions) runs on a second stream?
// Background: tools/ar_vs_gemm_neighbour.py - autoregressive decodes next to the bf16x3 GEMM (gemm_bx3.hip) differ from run to
// run, next to the fp32 GEMM they never do.
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_bf16_neighbour.hip -o mfma_bf16_neighbour && ./mfma_bf16_neighbour
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// ---- the neighbour: 256 threads, 56 KB of LDS, each wave streams fragments out of LDS through the matrix pipe
template <int MODE>   // 0: bf16 32x32x16, 1: fp32 32x32x2, 2: VALU only, 3: bf16 32x32x8 (older instruction), 4: bf16 16x16x32, 5: f16 32x32x16, 6: f16 16x16x32
__global__ __launch_bounds__(256) void neighbour(float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 57344 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = 1e-3f * (float)((i * 2654435761u) >> 20);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* base = lds + ((it & 1) * 28672) + (lane & 31) * 112 + 16 * (lane >> 5);
        bf16x8_t a[2][3], b[2][3];
        for (int i = 0; i < 2; ++i) for (int q = 0; q < 3; ++q) {
            a[i][q] = *reinterpret_cast<const bf16x8_t*>(base + i * 32 * 112 + 32 * q);
            b[i][q] = *reinterpret_cast<const bf16x8_t*>(base + (64 + i * 32) * 112 + 32 * q);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 3; ++q) {
            if (MODE == 0) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][q], b[j][2 - q], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2 - q], b[j][q], acc[i][j], 0, 0, 0);
            } else if (MODE == 3) {   // the older instruction (gfx90a+): K = 8 per issue, operands of 4 bf16
                typedef __attribute__((ext_vector_type(4))) short s16x4;
                const s16x4 alo = {(short)__builtin_bit_cast(unsigned short, a[i][q][0]), (short)__builtin_bit_cast(unsigned short, a[i][q][1]), (short)__builtin_bit_cast(unsigned short, a[i][q][2]), (short)__builtin_bit_cast(unsigned short, a[i][q][3])};
                const s16x4 blo = {(short)__builtin_bit_cast(unsigned short, b[j][q][0]), (short)__builtin_bit_cast(unsigned short, b[j][q][1]), (short)__builtin_bit_cast(unsigned short, b[j][q][2]), (short)__builtin_bit_cast(unsigned short, b[j][q][3])};
                const s16x4 ahi = {(short)__builtin_bit_cast(unsigned short, a[i][q][4]), (short)__builtin_bit_cast(unsigned short, a[i][q][5]), (short)__builtin_bit_cast(unsigned short, a[i][q][6]), (short)__builtin_bit_cast(unsigned short, a[i][q][7])};
                const s16x4 bhi = {(short)__builtin_bit_cast(unsigned short, b[j][q][4]), (short)__builtin_bit_cast(unsigned short, b[j][q][5]), (short)__builtin_bit_cast(unsigned short, b[j][q][6]), (short)__builtin_bit_cast(unsigned short, b[j][q][7])};
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(alo, blo, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ahi, bhi, acc[i][j], 0, 0, 0);
            } else if (MODE == 4) {   // the other double-rate bf16 shape of gfx950
                typedef __attribute__((ext_vector_type(4))) float f32x4;
                f32x4 t = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][q], b[j][2 - q], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2 - q], b[j][q], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][q], b[j][q], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2 - q], b[j][2 - q], t, 0, 0, 0);
                acc[i][j][0] = t[0]; acc[i][j][1] = t[1]; acc[i][j][2] = t[2]; acc[i][j][3] = t[3];
            } else if (MODE == 5 || MODE == 6) {   // the double-rate fp16 forms of gfx950 (same operand registers, read as 8 halves)
                typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
                const f16x8_t ha = __builtin_bit_cast(f16x8_t, a[i][q]), hb = __builtin_bit_cast(f16x8_t, b[j][2 - q]);
                if (MODE == 5) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc[i][j], 0, 0, 0);
                } else {
                    typedef __attribute__((ext_vector_type(4))) float f32x4;
                    f32x4 t = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, ha, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, hb, t, 0, 0, 0);
                    acc[i][j][0] = t[0]; acc[i][j][1] = t[1]; acc[i][j][2] = t[2]; acc[i][j][3] = t[3];
                }
            } else if (MODE == 1) {
                const float4 fa = __builtin_bit_cast(float4, a[i][q]), fb = __builtin_bit_cast(float4, b[j][q]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[i][j], 0, 0, 0);
            } else {
                const float4 fa = __builtin_bit_cast(float4, a[i][q]), fb = __builtin_bit_cast(float4, b[j][q]);
                acc[i][j][q] += fa.x * fb.y + fa.z * fb.w; acc[i][j][q + 3] += fa.y * fb.x + fa.w * fb.z;
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    if (s == 1.2345e-30f) sink[blockIdx.x * 256 + tid] = s;
}

// ---- the victim: 1024 threads, 54 KB of static LDS, weights held in registers across barriers, fp32 FMA chains, LDS sums
__global__ __launch_bounds__(1024) void victim(const float* __restrict__ w, const float* __restrict__ x, float* __restrict__ out) {
    __shared__ float4 part[16][64];
    __shared__ float4 part2[64][16];
    __shared__ float4 filler[1408];   // same LDS footprint as the step-tail kernel of the decoder (54 784 bytes)
    __shared__ __attribute__((aligned(16))) float xs[128];
    __shared__ __attribute__((aligned(16))) float h1[256];
    const unsigned tid = threadIdx.x, b = blockIdx.x;
    const unsigned kq = tid >> 6, j4 = tid & 63;
    float4 wv[5], w2[4];
    for (int i = 0; i < 5; ++i) wv[i] = reinterpret_cast<const float4*>(w)[(kq * 5 + i) * 64 + j4];
    for (int i = 0; i < 4; ++i) w2[i] = reinterpret_cast<const float4*>(w)[8192 + ((tid >> 4) * 4 + i) * 64 + (tid & 15)];
    if (tid < 128) xs[tid] = tid < 80 ? x[b * 80 + tid] : 0.f;
    if (tid == 0) filler[b & 1023 ? 0 : 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < 5; ++i) {
        const float v = xs[kq * 5 + i];
        acc.x = fmaf(wv[i].x, v, acc.x); acc.y = fmaf(wv[i].y, v, acc.y); acc.z = fmaf(wv[i].z, v, acc.z); acc.w = fmaf(wv[i].w, v, acc.w);
    }
    part[kq][j4] = acc;
    __syncthreads();
    if (tid < 256) {
        const float* col = reinterpret_cast<const float*>(part) + tid;
        float a = col[0];
        for (int q = 1; q < 16; ++q) a += col[q * 256];
        h1[tid] = fmaxf(a, 0.f);
    }
    __syncthreads();
    {
        const float4 h4 = reinterpret_cast<const float4*>(h1)[tid >> 4];
        float4 a2;
        a2.x = w2[0].x * h4.x; a2.y = w2[0].y * h4.x; a2.z = w2[0].z * h4.x; a2.w = w2[0].w * h4.x;
        a2.x = fmaf(w2[1].x, h4.y, a2.x); a2.y = fmaf(w2[1].y, h4.y, a2.y); a2.z = fmaf(w2[1].z, h4.y, a2.z); a2.w = fmaf(w2[1].w, h4.y, a2.w);
        a2.x = fmaf(w2[2].x, h4.z, a2.x); a2.y = fmaf(w2[2].y, h4.z, a2.y); a2.z = fmaf(w2[2].z, h4.z, a2.z); a2.w = fmaf(w2[2].w, h4.z, a2.w);
        a2.x = fmaf(w2[3].x, h4.w, a2.x); a2.y = fmaf(w2[3].y, h4.w, a2.y); a2.z = fmaf(w2[3].z, h4.w, a2.z); a2.w = fmaf(w2[3].w, h4.w, a2.w);
        part2[tid >> 4][tid & 15] = a2;
    }
    __syncthreads();
    if (tid < 64) {
        const float* col = reinterpret_cast<const float*>(part2) + tid;
        float a = col[0];
        for (int q = 1; q < 64; ++q) a += col[q * 64];
        out[b * 64 + tid] = a;
    }
}

// ---- register-only victims: one instruction kind each, 256 dependent operations per thread on values derived from the thread id
template <int OP>   // 0: v_fma_f32, 1: v_add_f32, 2: v_mul_f32, 3: v_pk_fma_f32 / packed if the compiler picks it (float2 fma)
__global__ __launch_bounds__(1024) void victim_reg(const float* __restrict__ w, float* __restrict__ out) {
    const unsigned tid = threadIdx.x, b = blockIdx.x;
    float a = w[(b * 1024 + tid) & 65535], c = w[(tid * 7 + b) & 65535] * 0.01f, r = 0.25f;
    float2 r2 = make_float2(0.25f, 0.5f);
#pragma unroll 16
    for (int i = 0; i < 256; ++i) {
        if (OP == 0) r = fmaf(r, a, c);
        else if (OP == 1) { r = r + a; a = a - c; }
        else if (OP == 2) { r = r * (1.f + c); }
        else { r2.x = fmaf(r2.x, a, c); r2.y = fmaf(r2.y, a, c); }
    }
    out[b * 1024 + tid] = OP == 3 ? r2.x + r2.y : r;
}

// ---- LDS-only victim: every thread publishes a value derived from its id, reads 16 values of other waves after a barrier, twice
template <int THREADS>
__global__ __launch_bounds__(THREADS) void victim_lds(float* __restrict__ out) {
    __shared__ float buf[2][1024];
    const unsigned tid = threadIdx.x, b = blockIdx.x;
    float v = 1.f + 1e-3f * (float)((tid * 2654435761u + b * 40503u) >> 22);
    for (int round = 0; round < 8; ++round) {
        buf[round & 1][tid] = v;
        __syncthreads();
        float a = 0.f;
        for (int q = 0; q < 16; ++q) a += buf[round & 1][(tid + 67 * q + 64) % THREADS];
        v = a * 0.0625f;
    }
    out[b * 1024 + tid] = v;
}

// ---- the victim's global loads alone (nine 16-byte loads per thread, summed in registers)
__global__ __launch_bounds__(1024) void victim_gl(const float* __restrict__ w, float* __restrict__ out) {
    const unsigned tid = threadIdx.x, b = blockIdx.x, kq = tid >> 6, j4 = tid & 63;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < 5; ++i) { const float4 v = reinterpret_cast<const float4*>(w)[(kq * 5 + i) * 64 + j4]; s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; }
    for (int i = 0; i < 4; ++i) { const float4 v = reinterpret_cast<const float4*>(w)[8192 + ((tid >> 4) * 4 + i) * 64 + (tid & 15)]; s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; }
    out[b * 1024 + tid] = (s4.x + s4.y) + (s4.z + s4.w) + (float)b;
}
// ---- 16-byte LDS traffic alone in the victim's footprint (54 KB static, 1024 threads)
__global__ __launch_bounds__(1024) void victim_l128(float* __restrict__ out) {
    __shared__ float4 part[16][64];
    __shared__ float4 part2[64][16];
    __shared__ float4 filler[1408];
    const unsigned tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) filler[b & 1023 ? 0 : 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    float v = 1.f + 1e-3f * (float)((tid * 2654435761u + b * 40503u) >> 22);
    for (int round = 0; round < 4; ++round) {
        part[tid >> 6][tid & 63] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
        __syncthreads();
        const float4 p = part[(tid + 5) & 15][(tid * 7 + 3) & 63];
        part2[tid >> 4][tid & 15] = make_float4(p.w, p.z, p.y, p.x);
        __syncthreads();
        const float4 q = part2[(tid + 9) & 63][(tid >> 6) & 15];
        v = 0.25f * ((q.x + q.y) + (q.z + q.w));
        __syncthreads();
    }
    out[b * 1024 + tid] = v;
}

// ---- LDS reads where many lanes share an address (the victim's x / h1 reads), fp32 FMAs with per-thread constants, LDS column sums
template <int WIDE>   // 0: 4-byte reads, one address per wave; 1: 16-byte reads, one address per 16 lanes
__global__ __launch_bounds__(1024) void victim_bcast(float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float xs[256];
    __shared__ float part[16][1024 / 16];
    const unsigned tid = threadIdx.x, b = blockIdx.x, kq = tid >> 6;
    if (tid < 256) xs[tid] = 0.5f + 1e-3f * (float)((tid * 2654435761u + b * 40503u) >> 22);
    const float c = 1.f + 1e-3f * (float)(tid & 127);
    __syncthreads();
    float acc = 0.f;
    if (WIDE == 0) {
        for (int i = 0; i < 16; ++i) acc = fmaf(c, xs[kq * 16 + i], acc);
    } else {
        for (int i = 0; i < 4; ++i) {
            const float4 v = reinterpret_cast<const float4*>(xs)[((tid >> 4) + 16 * i) & 63];
            acc = fmaf(c, v.x, acc); acc = fmaf(c, v.y, acc); acc = fmaf(c, v.z, acc); acc = fmaf(c, v.w, acc);
        }
    }
    part[kq][tid & 63] = acc;
    __syncthreads();
    if (tid < 64) {
        float a = part[0][tid];
        for (int q = 1; q < 16; ++q) a += part[q][tid];
        out[b * 1024 + tid] = a;
    }
}

__global__ void compare(const unsigned* a, const unsigned* ref, int n, unsigned* mismatches) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != ref[i]) atomicAdd(mismatches, 1u);
}

template <int MODE, int OP> static int trial_reg(const char* name, hipStream_t sv, hipStream_t sn, const float* w, float* out, float* ref, float* sink,
                                                 unsigned* mism, int rounds) {
    const int n = 128 * 1024;
    victim_reg<OP><<<128, 1024, 0, sv>>>(w, ref);
    CK(hipDeviceSynchronize());
    CK(hipMemset(mism, 0, 4));
    for (int r = 0; r < rounds; ++r) {
        if ((r % 64) == 0) neighbour<MODE><<<800, 256, 57344, sn>>>(sink, 1500);
        victim_reg<OP><<<128, 1024, 0, sv>>>(w, out);
        compare<<<(n + 255) / 256, 256, 0, sv>>>(reinterpret_cast<const unsigned*>(out), reinterpret_cast<const unsigned*>(ref), n, mism);
    }
    CK(hipDeviceSynchronize());
    unsigned h; CK(hipMemcpy(&h, mism, 4, hipMemcpyDeviceToHost));
    printf("%-58s %6d launches: %u mismatching words\n", name, rounds, h);
    return 0;
}

template <int MODE, int THREADS> static int trial_lds(const char* name, hipStream_t sv, hipStream_t sn, float* out, float* ref, float* sink, unsigned* mism, int rounds) {
    const int n = 128 * 1024;
    CK(hipMemset(out, 0, n * 4)); CK(hipMemset(ref, 0, n * 4));
    victim_lds<THREADS><<<128, THREADS, 0, sv>>>(ref);
    CK(hipDeviceSynchronize());
    CK(hipMemset(mism, 0, 4));
    for (int r = 0; r < rounds; ++r) {
        if ((r % 64) == 0) neighbour<MODE><<<800, 256, 57344, sn>>>(sink, 1500);
        victim_lds<THREADS><<<128, THREADS, 0, sv>>>(out);
        compare<<<(n + 255) / 256, 256, 0, sv>>>(reinterpret_cast<const unsigned*>(out), reinterpret_cast<const unsigned*>(ref), n, mism);
    }
    CK(hipDeviceSynchronize());
    unsigned h; CK(hipMemcpy(&h, mism, 4, hipMemcpyDeviceToHost));
    printf("%-58s %6d launches: %u mismatching words\n", name, rounds, h);
    return 0;
}

template <int MODE, int WHICH> static int trial_part(const char* name, hipStream_t sv, hipStream_t sn, const float* w, float* out, float* ref, float* sink, unsigned* mism, int rounds) {
    const int n = 128 * 1024;
    if (WHICH == 0) victim_gl<<<128, 1024, 0, sv>>>(w, ref); else victim_l128<<<128, 1024, 0, sv>>>(ref);
    CK(hipDeviceSynchronize());
    CK(hipMemset(mism, 0, 4));
    for (int r = 0; r < rounds; ++r) {
        if ((r % 64) == 0) neighbour<MODE><<<800, 256, 57344, sn>>>(sink, 1500);
        if (WHICH == 0) victim_gl<<<128, 1024, 0, sv>>>(w, out); else victim_l128<<<128, 1024, 0, sv>>>(out);
        compare<<<(n + 255) / 256, 256, 0, sv>>>(reinterpret_cast<const unsigned*>(out), reinterpret_cast<const unsigned*>(ref), n, mism);
    }
    CK(hipDeviceSynchronize());
    unsigned h; CK(hipMemcpy(&h, mism, 4, hipMemcpyDeviceToHost));
    printf("%-58s %6d launches: %u mismatching words\n", name, rounds, h);
    return 0;
}

template <int MODE, int WIDE> static int trial_bcast(const char* name, hipStream_t sv, hipStream_t sn, float* out, float* ref, float* sink, unsigned* mism, int rounds) {
    const int n = 128 * 1024;
    CK(hipMemset(out, 0, n * 4)); CK(hipMemset(ref, 0, n * 4));
    victim_bcast<WIDE><<<128, 1024, 0, sv>>>(ref);
    CK(hipDeviceSynchronize());
    CK(hipMemset(mism, 0, 4));
    for (int r = 0; r < rounds; ++r) {
        if ((r % 64) == 0) neighbour<MODE><<<800, 256, 57344, sn>>>(sink, 1500);
        victim_bcast<WIDE><<<128, 1024, 0, sv>>>(out);
        compare<<<(n + 255) / 256, 256, 0, sv>>>(reinterpret_cast<const unsigned*>(out), reinterpret_cast<const unsigned*>(ref), n, mism);
    }
    CK(hipDeviceSynchronize());
    unsigned h; CK(hipMemcpy(&h, mism, 4, hipMemcpyDeviceToHost));
    printf("%-58s %6d launches: %u mismatching words\n", name, rounds, h);
    return 0;
}

template <int MODE> static int trial(const char* name, hipStream_t sv, hipStream_t sn, const float* w, const float* x, float* out, const float* ref,
                                     float* sink, unsigned* mism, int rounds) {
    const int WGS = 128, n = WGS * 64;
    CK(hipMemset(mism, 0, 4));
    int bad_rounds = 0;
    unsigned prev = 0;
    for (int r = 0; r < rounds; ++r) {
        if (MODE >= 0 && (r % 64) == 0) neighbour<(MODE < 0 ? 2 : MODE)><<<800, 256, 57344, sn>>>(sink, 1500);   // ~ms each: always one in flight
        victim<<<WGS, 1024, 0, sv>>>(w, x, out);
        compare<<<(n + 255) / 256, 256, 0, sv>>>(reinterpret_cast<const unsigned*>(out), reinterpret_cast<const unsigned*>(ref), n, mism);
        if ((r % 64) == 63) {
            CK(hipStreamSynchronize(sv));
            unsigned h; CK(hipMemcpy(&h, mism, 4, hipMemcpyDeviceToHost));
            if (h != prev) ++bad_rounds;
            prev = h;
        }
    }
    CK(hipDeviceSynchronize());
    unsigned h; CK(hipMemcpy(&h, mism, 4, hipMemcpyDeviceToHost));
    printf("%-34s %6d victim launches: %u mismatching words (in %d of %d groups of 64 launches)\n", name, rounds, h, bad_rounds, rounds / 64);
    return 0;
}

// which XCDs a stream's workgroups land on (CU-masked streams: MASK=... below)
__global__ void where_kernel(unsigned* xcc_bits) {
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicOr(xcc_bits, 1u << (xcc & 0xf));
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 4096;
    // MASK=halves | xcds | same: victim and neighbour on CU-masked streams (hipExtStreamCreateWithCUMask) -
    //   halves: CU bits [0, 128) / [128, 256);  xcds: bits with (i % 8) < 4 / >= 4;  same: both on bits [0, 128) (control)
    // Round 4: is the disturbance a per-CU matter (it vanishes on disjoint CUs) or chip-wide?
    const char* mask_mode = getenv("MASK");
    if (mask_mode) {
        hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
        const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
        std::vector<uint32_t> mv(words, 0), mn(words, 0);
        for (int i = 0; i < ncu; ++i) {
            bool v, n;
            if (mask_mode[0] == 'h') { v = i < ncu / 2; n = !v; }
            else if (mask_mode[0] == 'x') { v = (i % 8) < 4; n = !v; }
            else { v = i < ncu / 2; n = v; }
            if (v) mv[i / 32] |= 1u << (i % 32);
            if (n) mn[i / 32] |= 1u << (i % 32);
        }
        for (int m = 0; m < 7; ++m) {
            const void* f = m == 0 ? (const void*)neighbour<0> : m == 1 ? (const void*)neighbour<1> : m == 2 ? (const void*)neighbour<2> : m == 3 ? (const void*)neighbour<3>
                          : m == 4 ? (const void*)neighbour<4> : m == 5 ? (const void*)neighbour<5> : (const void*)neighbour<6>;
            CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        std::vector<float> hw(8192 * 4 + 65536 * 4), hx(128 * 80);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
        for (auto& v : hw) v = rnd();
        for (auto& v : hx) v = rnd();
        float *w, *x, *out, *ref, *sink; unsigned* mism; unsigned* bits;
        CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&out, 128 * 64 * 4)); CK(hipMalloc(&ref, 128 * 64 * 4));
        CK(hipMalloc(&sink, 800 * 256 * 4)); CK(hipMalloc(&mism, 4)); CK(hipMalloc(&bits, 8)); CK(hipMemset(bits, 0, 8));
        CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        hipStream_t sv, sn;
        CK(hipExtStreamCreateWithCUMask(&sv, words, mv.data())); CK(hipExtStreamCreateWithCUMask(&sn, words, mn.data()));
        where_kernel<<<2048, 64, 0, sv>>>(bits); where_kernel<<<2048, 64, 0, sn>>>(bits + 1);
        CK(hipDeviceSynchronize());
        unsigned hb[2]; CK(hipMemcpy(hb, bits, 8, hipMemcpyDeviceToHost));
        printf("MASK=%s: victim stream ran on XCDs 0x%02x, neighbour stream on XCDs 0x%02x\n", mask_mode, hb[0], hb[1]);
        victim<<<128, 1024, 0, sv>>>(w, x, ref);
        CK(hipDeviceSynchronize());
        if (trial<-1>("alone", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
        if (trial<1>("next to v_mfma_f32_32x32x2_f32", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
        if (trial<0>("next to v_mfma_f32_32x32x16_bf16", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
        if (trial<4>("next to v_mfma_f32_16x16x32_bf16", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
        if (trial<6>("next to v_mfma_f32_16x16x32_f16", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
        return 0;
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(neighbour<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(neighbour<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(neighbour<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(neighbour<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(neighbour<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(neighbour<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(neighbour<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<float> hw(8192 * 4 + 65536 * 4), hx(128 * 80);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hw) v = rnd();
    for (auto& v : hx) v = rnd();
    float *w, *x, *out, *ref, *sink; unsigned* mism;
    CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&out, 128 * 64 * 4)); CK(hipMalloc(&ref, 128 * 64 * 4));
    CK(hipMalloc(&sink, 800 * 256 * 4)); CK(hipMalloc(&mism, 4));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sn;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sn, hipStreamNonBlocking));
    victim<<<128, 1024, 0, sv>>>(w, x, ref);
    CK(hipDeviceSynchronize());
    if (trial<-1>("alone", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
    if (trial<2>("next to VALU + LDS", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
    if (trial<1>("next to v_mfma_f32_32x32x2_f32", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
    if (trial<0>("next to v_mfma_f32_32x32x16_bf16", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
    if (trial<1>("next to v_mfma_f32_32x32x2_f32", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
    if (trial<0>("next to v_mfma_f32_32x32x16_bf16", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
    float *outr, *refr;
    CK(hipMalloc(&outr, 128 * 1024 * 4)); CK(hipMalloc(&refr, 128 * 1024 * 4));
    if (trial_reg<1, 0>("registers only, v_fma_f32 chain, next to fp32 mfma", sv, sn, w, outr, refr, sink, mism, rounds)) return 1;
    if (trial_reg<0, 0>("registers only, v_fma_f32 chain, next to bf16 mfma", sv, sn, w, outr, refr, sink, mism, rounds)) return 1;
    if (trial_reg<0, 1>("registers only, v_add_f32 chain, next to bf16 mfma", sv, sn, w, outr, refr, sink, mism, rounds)) return 1;
    if (trial_reg<0, 2>("registers only, v_mul_f32 chain, next to bf16 mfma", sv, sn, w, outr, refr, sink, mism, rounds)) return 1;
    if (trial_reg<0, 3>("registers only, two fma chains, next to bf16 mfma", sv, sn, w, outr, refr, sink, mism, rounds)) return 1;
    if (trial_lds<1, 1024>("LDS exchange only, 1024 threads, next to fp32 mfma", sv, sn, outr, refr, sink, mism, rounds)) return 1;
    if (trial_lds<0, 1024>("LDS exchange only, 1024 threads, next to bf16 mfma", sv, sn, outr, refr, sink, mism, rounds)) return 1;
    if (trial_lds<0, 256>("LDS exchange only, 256 threads, next to bf16 mfma", sv, sn, outr, refr, sink, mism, rounds)) return 1;
    if (trial_lds<0, 64>("LDS exchange only, 64 threads (one wave), next to bf16 mfma", sv, sn, outr, refr, sink, mism, rounds)) return 1;
    if (trial<3>("next to v_mfma_f32_32x32x8_bf16_1k", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
    if (trial<4>("next to v_mfma_f32_16x16x32_bf16", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
    if (trial<5>("next to v_mfma_f32_32x32x16_f16", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
    if (trial<6>("next to v_mfma_f32_16x16x32_f16", sv, sn, w, x, out, ref, sink, mism, rounds)) return 1;
    if (trial_part<0, 0>("global 16-byte loads only, next to bf16 32x32x16", sv, sn, w, outr, refr, sink, mism, rounds)) return 1;
    if (trial_part<0, 1>("LDS 16-byte exchange in 54 KB, next to bf16 32x32x16", sv, sn, w, outr, refr, sink, mism, rounds)) return 1;
    if (trial_bcast<4, 0>("LDS 4-byte broadcast reads + fma, next to bf16 16x16x32", sv, sn, outr, refr, sink, mism, rounds)) return 1;
    if (trial_bcast<4, 1>("LDS 16-byte shared reads + fma, next to bf16 16x16x32", sv, sn, outr, refr, sink, mism, rounds)) return 1;
    if (trial_bcast<1, 1>("LDS 16-byte shared reads + fma, next to fp32 mfma", sv, sn, outr, refr, sink, mism, rounds)) return 1;
    {   // a few mismatching words of the full victim next to the bf16 neighbour
        std::vector<float> hr(128 * 64), ho(128 * 64);
        CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
        int shown = 0;
        for (int r = 0; r < 2048 && shown < 12; ++r) {
            if ((r % 64) == 0) neighbour<0><<<800, 256, 57344, sn>>>(sink, 1500);
            victim<<<128, 1024, 0, sv>>>(w, x, out);
            CK(hipStreamSynchronize(sv));
            CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ho.size() && shown < 12; ++i)
                if (ho[i] != hr[i]) { printf("  launch %d workgroup %zu column %zu: %.9g instead of %.9g\n", r, i / 64, i % 64, ho[i], hr[i]); ++shown; }
        }
        CK(hipDeviceSynchronize());
    }
    return 0;
}
