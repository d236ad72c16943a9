// Practical ceiling of v_mfma_f32_32x32x16_bf16 on the whole chip: register-only MFMA chains (no LDS, no memory),
// NACC independent accumulators per wave, WPS waves per SIMD.  Prints TF/s for a few (NACC, WPS) pairs so that the conv
// kernels' "fraction of peak" can be read against what the matrix pipe sustains under its own power limit.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe scripts/probes/mfma_peak_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, const bf16x8* rnd) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f - threadIdx.x * 0.002f); }
    if (rnd) { a = rnd[threadIdx.x]; b = rnd[512 + threadIdx.x]; }      // operands with random mantissas: the data-dependent power
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// The conv kernels' inner loop in miniature: RD ds_read_b128 fragment reads (conflict-free, random bf16 data in LDS) feeding
// 6 MFMAs on 2 accumulators, 2 waves per SIMD, no barriers, no global traffic — what the matrix pipe sustains when the LDS
// is busy at the conv kernels' rate (RD = 7: the 256-voxel kernel's 1.17 reads per MFMA; RD = 4: 0.69 per MFMA).
template <int RD>
__global__ __launch_bounds__(512) void k_lds(float* out, int iters, const bf16x8* rnd) {
    __shared__ bf16x8 img[8192];                                   // 128 KB
    for (int i = threadIdx.x; i < 8192; i += 512) img[i] = rnd[i & 1023];
    __syncthreads();
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    int base = threadIdx.x & 63;                                   // 64 lanes x 16 B = 1 KB contiguous: all banks, no conflicts
    bf16x8 f[2][7];
    for (int r = 0; r < RD; ++r) f[0][r] = img[(base + 64 * (r + 8 * (threadIdx.x >> 6))) & 8191];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < RD; ++r) f[h ^ 1][r] = img[(base + 64 * (r + 7 * it + 8 * (threadIdx.x >> 6) + 3 * h)) & 8191];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[h][t % RD], f[h][(t + 1) % RD], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[h][(t + 2) % RD], f[h][(t + 3) % RD], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int RD>
void run_lds(int iters, float* d, const bf16x8* rnd) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_lds<RD>, dim3(256), dim3(512), 0, 0, d, iters, rnd);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_lds<RD>, dim3(256), dim3(512), 0, 0, d, iters, rnd);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 256.0 * 8 * (double)iters * 12 * 32768.0;
    printf("MFMA + %d ds_read_b128 per 6 MFMAs (random operands from LDS, 8 waves/CU): %.3f ms  %.0f TF/s\n", RD, best, flop / best / 1e9);
}

template <int NACC>
void run(int threads, int iters, float* d, const bf16x8* rnd = nullptr) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) (void)hipGetLastError(); hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, d, iters, rnd);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        (void)hipGetLastError(); hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, d, iters, rnd);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 256.0 * (threads / 64) * (double)iters * 4 * NACC * 32 * 32 * 16 * 2;
    printf("%s NACC %d, %d waves/CU (%d per SIMD): %.3f ms  %.0f TF/s\n", rnd ? "random operands  " : "constant operands", NACC, threads / 64, threads / 256, best, flop / best / 1e9);
}

int main() {
    float* d; hipMalloc(&d, 4096);
    const int it = 20000;
    run<1>(256, it, d); run<2>(256, it, d); run<4>(256, it, d); run<8>(256, it / 2, d);
    run<1>(512, it, d); run<2>(512, it, d); run<4>(512, it / 2, d);
    {
        unsigned short* h = new unsigned short[1024 * 8];
        unsigned long long x = 88172645463325252ull;
        for (int i = 0; i < 1024 * 8; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (unsigned short)(0x3c00 | (x & 0x83ff)); }   // |v| in [0.0078, 0.0156), random sign + mantissa
        bf16x8* r; (void)hipMalloc(&r, 1024 * 16); (void)hipMemcpy(r, h, 1024 * 16, hipMemcpyHostToDevice);
        run<2>(512, it, d, r); run<4>(256, it, d, r);
        run_lds<7>(it / 4, d, r); run_lds<4>(it / 4, d, r); run_lds<2>(it / 4, d, r);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        for (int q = 0; q < 40; ++q) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d, it, r);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("random operands, sustained (40 launches, NACC 2, 8 waves/CU): %.1f ms  %.0f TF/s\n", ms, 40.0 * 256 * 8 * it * 4 * 2 * 32768.0 / ms / 1e9);
    }
    // long run: what the pipe sustains once the power limit has pulled the clock down
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d, it, (const bf16x8*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("sustained (40 launches, NACC 2, 8 waves/CU): %.1f ms  %.0f TF/s\n", ms, 40.0 * 256 * 8 * it * 4 * 2 * 32768.0 / ms / 1e9);
    return 0;
}
