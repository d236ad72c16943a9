// VALU issue-cost probe for gfx950: cycles per wave-instruction of the opcodes the attention soft-max uses, measured with
// s_memtime around an unrolled block of INDEPENDENT instructions (8 chains), for 1 / 2 / 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_issue_probe valu_issue_probe.hip ; run: ./valu_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int OP>
__global__ void probe(unsigned long long* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 0.999f;
    const v2f c2 = {c, c};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = a0 + r; acc1[r] = a1 - r; }
    bf16x8 fa, fb;
    for (int r = 0; r < 8; ++r) { fa[r] = (__bf16)(a2 * 1e-3f + r); fb[r] = (__bf16)(a3 * 1e-3f - r); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 16; ++it) {
        if constexpr (OP == 0) {   // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if constexpr (OP == 1) {   // v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                              "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if constexpr (OP == 2) {   // v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (OP == 3) {   // v_cvt_pk_bf16_f32
            REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                              "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (OP == 4) {   // v_max3_f32
            REP8(asm volatile("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n"
                              "v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if constexpr (OP == 5) {   // v_pk_add_f32
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if constexpr (OP == 6) {   // v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if constexpr (OP == 7) {   // dependent chain of v_fma_f32 (latency)
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(c));)
        } else if constexpr (OP == 8) {   // dependent chain of v_exp_f32
            REP64(asm volatile("v_exp_f32 %0, %0" : "+v"(a0));)
        } else if constexpr (OP == 9) {   // dependent chain of v_pk_add_f32
            REP64(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(c2));)
        } else if constexpr (OP == 10) {  // dependent chain of v_max3_f32
            REP64(asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(c));)
        } else if constexpr (OP == 11) {  // v_exp_f32 and v_fma_f32 alternating (do the transcendental and the plain pipe overlap?)
            REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %8\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %8, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if constexpr (OP == 12) {  // v_cvt_pk_bf16_f32 dependent chain
            REP64(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a0));)
        } else if constexpr (OP == 13) {  // 8 MFMAs on two independent accumulators, nothing else
            REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n"
                              "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n"
                              "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n"
                              "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1"
                              : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb));)
        } else if constexpr (OP == 14 || OP == 15 || OP == 16) {  // per MFMA: 6 independent v_fma_f32 (14), 4 v_fma + 2 v_exp (15), 8 v_fma (16)
#define GRP14(ACC) "v_mfma_f32_32x32x16_bf16 " ACC ", %10, %11, " ACC "\n v_fma_f32 %0, %0, %12, %12\n v_fma_f32 %1, %1, %12, %12\n v_fma_f32 %2, %2, %12, %12\n" \
                   "v_fma_f32 %3, %3, %12, %12\n v_fma_f32 %4, %4, %12, %12\n v_fma_f32 %5, %5, %12, %12\n"
#define GRP15(ACC) "v_mfma_f32_32x32x16_bf16 " ACC ", %10, %11, " ACC "\n v_fma_f32 %0, %0, %12, %12\n v_fma_f32 %1, %1, %12, %12\n v_exp_f32 %2, %2\n" \
                   "v_fma_f32 %3, %3, %12, %12\n v_fma_f32 %4, %4, %12, %12\n v_exp_f32 %5, %5\n"
#define GRP16(ACC) "v_mfma_f32_32x32x16_bf16 " ACC ", %10, %11, " ACC "\n v_fma_f32 %0, %0, %12, %12\n v_fma_f32 %1, %1, %12, %12\n v_fma_f32 %2, %2, %12, %12\n" \
                   "v_fma_f32 %3, %3, %12, %12\n v_fma_f32 %4, %4, %12, %12\n v_fma_f32 %5, %5, %12, %12\n v_fma_f32 %6, %6, %12, %12\n v_fma_f32 %7, %7, %12, %12\n"
            if constexpr (OP == 14) {
                REP8(asm volatile(GRP14("%8") GRP14("%9") GRP14("%8") GRP14("%9") GRP14("%8") GRP14("%9") GRP14("%8") GRP14("%9")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(acc0), "+v"(acc1)
                                  : "v"(fa), "v"(fb), "v"(c));)
            } else if constexpr (OP == 15) {
                REP8(asm volatile(GRP15("%8") GRP15("%9") GRP15("%8") GRP15("%9") GRP15("%8") GRP15("%9") GRP15("%8") GRP15("%9")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(acc0), "+v"(acc1)
                                  : "v"(fa), "v"(fb), "v"(c));)
            } else {
                REP8(asm volatile(GRP16("%8") GRP16("%9") GRP16("%8") GRP16("%9") GRP16("%8") GRP16("%9") GRP16("%8") GRP16("%9")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(acc0), "+v"(acc1)
                                  : "v"(fa), "v"(fb), "v"(c));)
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (OP >= 13) a0 += acc0[3] + acc1[7];
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.y + p3.x + p4.x + p5.y + p6.x + p7.x == 12345.678f) out[0] = 0;
}

template <int OP>
void run(const char* name, int per_iter) {
    unsigned long long* d;
    hipMalloc(&d, 256 * 16 * sizeof(unsigned long long));
    printf("%-44s", name);
    for (int waves : {4, 8, 16}) {                 // waves per workgroup = 1 / 2 / 4 per SIMD, one workgroup per CU
        hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(64 * waves), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(64 * waves), 0, 0, d, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us_per_launch = ms * 1e3 / 20;
        std::vector<unsigned long long> h(256 * waves);
        hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        const double per_wave_instr = s / h.size() / (16.0 * per_iter);
        printf("  %d/SIMD: %6.2f cyc/instr/wave (%5.2f per SIMD; %.0f ticks in %.1f us)", waves / 4, per_wave_instr, per_wave_instr / (waves / 4), s / h.size(), us_per_launch);
    }
    printf("\n");
    hipFree(d);
}

int main() {
    printf("cycles (s_memtime ticks) per wave-instruction; 'per SIMD' = issue interval of the SIMD's VALU with that many waves\n");
    run<0>("v_fma_f32 (8 independent chains)", 64);
    run<6>("v_mul_f32", 64);
    run<1>("v_pk_fma_f32", 64);
    run<5>("v_pk_add_f32", 64);
    run<2>("v_exp_f32", 64);
    run<3>("v_cvt_pk_bf16_f32", 64);
    run<4>("v_max3_f32", 64);
    run<11>("v_exp_f32 / v_fma_f32 alternating", 64);
    run<7>("v_fma_f32 dependent chain", 64);
    run<8>("v_exp_f32 dependent chain", 64);
    run<9>("v_pk_add_f32 dependent chain", 64);
    run<10>("v_max3_f32 dependent chain", 64);
    run<12>("v_cvt_pk_bf16_f32 dependent chain", 64);
    printf("MFMA groups: cycles per GROUP (= per MFMA)\n");
    run<13>("v_mfma_f32_32x32x16_bf16 alone", 64);
    run<14>("1 MFMA + 6 v_fma_f32", 64);
    run<16>("1 MFMA + 8 v_fma_f32", 64);
    run<15>("1 MFMA + 4 v_fma_f32 + 2 v_exp_f32", 64);
    return 0;
}
