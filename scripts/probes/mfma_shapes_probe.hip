// Which matrix-instruction SHAPE does the most work per joule?  At the chip's power limit the sustained rate of a register-only chain
// of random-operand MFMAs is a proxy for energy per flop.  32x32x16 vs 16x16x32 (bf16), the fp8 forms, the block-scaled 32x32x64.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shapes_probe scripts/probes/mfma_shapes_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int iters, const int* rnd) {
    const int l = threadIdx.x;
    i32x8 ra, rb;
    for (int i = 0; i < 8; ++i) { ra[i] = rnd[(l * 8 + i) & 4095]; rb[i] = rnd[(l * 8 + i + 977) & 4095]; }
    bf16x8 a = __builtin_bit_cast(bf16x8, (int __attribute__((ext_vector_type(4)))){ra[0], ra[1], ra[2], ra[3]});
    bf16x8 b = __builtin_bit_cast(bf16x8, (int __attribute__((ext_vector_type(4)))){rb[0], rb[1], rb[2], rb[3]});
    float s = 0.f;
    if constexpr (SHAPE == 0) {            // 32x32x16 bf16, two accumulators
        f32x16 c0, c1;
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    } else if constexpr (SHAPE == 1) {     // 16x16x32 bf16, four accumulators (same flops per iteration: 16 x 16 K)
        f32x4 c[4];
        for (int n = 0; n < 4; ++n) c[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int n = 0; n < 4; ++n) c[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((n & 1) ? b : a, (n & 1) ? a : b, c[n], 0, 0, 0);
            }
        }
        for (int n = 0; n < 4; ++n) s += c[n][0] + c[n][1] + c[n][2] + c[n][3];
    } else if constexpr (SHAPE == 2) {     // 32x32x64 block-scaled e4m3 (run-time scales 127), two accumulators
        f32x16 c0, c1;
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
        const int sc = 127 + (rnd[l & 4095] & 0);
        for (int i = 0; i < 8; ++i) { ra[i] &= 0x77777777; rb[i] &= 0x77777777; }      // finite e4m3 values
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ra, rb, c0, 0, 0, 0, sc, 0, sc);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(rb, ra, c1, 0, 0, 0, sc, 0, sc);
            }
        }
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    } else {                               // 32x32x16 fp8 (non-scaled), two accumulators
        f32x16 c0, c1;
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
        const long la = ((long)(ra[0] & 0x77777777) << 32) | (unsigned)(ra[1] & 0x77777777), lb = ((long)(rb[0] & 0x77777777) << 32) | (unsigned)(rb[1] & 0x77777777);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(la, lb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(lb, la, c1, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    }
    if (s == 123.456f) out[l] = s;
}

// the conv kernels' inner loop in miniature (as scripts/probes/mfma_peak_probe.hip: 7 conflict-free ds_read_b128 fragment reads per
// 6 x 32x32x16 MFMAs, 2 waves per SIMD, no barriers) — and the same bytes and flops on 12 x 16x16x32 MFMAs with eight 16 x 16 accumulators
template <int SMALL>
__global__ __launch_bounds__(512) void k_lds(float* out, int iters, const int* rnd) {
    __shared__ bf16x8 img[8192];                                   // 128 KB
    for (int i = threadIdx.x; i < 8192; i += 512) {
        const int* r4 = rnd + ((i * 4) & 4095);
        img[i] = __builtin_bit_cast(bf16x8, (int __attribute__((ext_vector_type(4)))){r4[0], r4[1], r4[2], r4[3]});
    }
    __syncthreads();
    const int base = threadIdx.x & 63;
    bf16x8 f[2][7];
    for (int r = 0; r < 7; ++r) f[0][r] = img[(base + 64 * (r + 8 * (threadIdx.x >> 6))) & 8191];
    float s = 0.f;
    if constexpr (!SMALL) {
        f32x16 acc[2];
        for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 7; ++r) f[h ^ 1][r] = img[(base + 64 * (r + 7 * it + 8 * (threadIdx.x >> 6) + 3 * h)) & 8191];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[h][t], f[h][t + 1], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[h][t + 4], f[h][t + 3], acc[1], 0, 0, 0);
                }
#pragma unroll
                for (int i_ = 0; i_ < 6; ++i_) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    } else {
        f32x4 acc[8];
        for (int n = 0; n < 8; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 7; ++r) f[h ^ 1][r] = img[(base + 64 * (r + 7 * it + 8 * (threadIdx.x >> 6) + 3 * h)) & 8191];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[(t & 1) * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[h][t + (n >> 1) * 3], f[h][t + 1 + (n & 1) * 2], acc[(t & 1) * 4 + n], 0, 0, 0);
                }
#pragma unroll
                for (int i_ = 0; i_ < 6; ++i_) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); }
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int n = 0; n < 8; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int SMALL>
static void run_lds(const char* name, const int* rnd, float* out) {
    const int iters = 6000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        for (int n = 0; n < 5; ++n) k_lds<SMALL><<<256, 512>>>(out, iters, rnd);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    const double flop = 5.0 * 256 * 8 * iters * 12 * 32768.0;
    printf("%-46s %8.2f ms for 5 launches  %6.0f TF/s\n", name, best, flop / best / 1e9);
}

template <int SHAPE>
static void run(const char* name, double flop_per_iter_per_wave, const int* rnd, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        for (int n = 0; n < 5; ++n) k<SHAPE><<<256, 512>>>(out, iters, rnd);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    const double flop = 5.0 * 256 * 8 * iters * flop_per_iter_per_wave;
    printf("%-46s %8.2f ms for 5 launches  %6.0f TF/s (sustained over >= 100 ms)\n", name, best, flop / best / 1e9);
}

int main() {
    std::vector<int> h(4096);
    std::mt19937 g(7);
    for (auto& v : h) { v = (int)g(); v &= 0x3fff3fff; v |= 0x3c003c00; }      // bf16 pairs in [0.5, 2): finite, random mantissas / signs off
    int* rnd; float* out;
    hipMalloc(&rnd, 4096 * 4); hipMalloc(&out, 4096);
    hipMemcpy(rnd, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0>("v_mfma_f32_32x32x16_bf16 (2 accumulators)", 8 * 32768.0, rnd, out);
    run<1>("v_mfma_f32_16x16x32_bf16 (4 accumulators)", 16 * 16384.0, rnd, out);
    run<3>("v_mfma_f32_32x32x16_fp8_fp8 (2 accumulators)", 8 * 32768.0, rnd, out);
    run<2>("v_mfma_scale_f32_32x32x64_f8f6f4 e4m3", 4 * 131072.0, rnd, out);
    run<0>("v_mfma_f32_32x32x16_bf16 again", 8 * 32768.0, rnd, out);
    run_lds<0>("7 ds_read_b128 + 6 x 32x32x16 (LDS-fed)", rnd, out);
    run_lds<1>("7 ds_read_b128 + 12 x 16x16x32 (LDS-fed)", rnd, out);
    run_lds<0>("7 ds_read_b128 + 6 x 32x32x16 (LDS-fed) again", rnd, out);
    return 0;
}
