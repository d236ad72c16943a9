// Which packed-fp32 operand form is corrupted beside hupr_k_conv_halo_bf16<64, 64>?  (round 4)
// Each thread evaluates ONE instruction form on fixed operands `iters` x 8 times and xors the result bits of consecutive evaluations:
// every evaluation must give the same bits, so the accumulated xor is 0 unless some evaluation returned something else.
// build (GPU box): hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libpk_victim.so scripts/probes/pk_victim.hip
#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int FORM>
__global__ __launch_bounds__(256) void pk_victim(const v2f* __restrict__ in, unsigned* __restrict__ bad, int iters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    v2f a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = in[(i * 16 + k) & 0xfffff]; b[k] = in[(i * 16 + 8 + k) & 0xfffff]; }
    unsigned long long acc = 0;
    const v2f pm = {1.0f, -1.0f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v2f d;
            if constexpr (FORM == 0) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 1) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 5) { asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(d.x) : "v"(a[k].x), "v"(b[k].x)); d.y = d.x; }
            else if constexpr (FORM == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %2" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 7) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 8) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 9) asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 10) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 11) asm volatile("v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a[k]), "v"(b[k]), "s"(pm));
            else if constexpr (FORM == 12) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 13) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 14) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 15) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 16) asm volatile("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            else if constexpr (FORM == 17) asm volatile("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(d) : "v"(a[k]), "v"(b[k]));
            const unsigned long long bits = ((unsigned long long)__float_as_uint(d.y) << 32) | __float_as_uint(d.x);
            acc ^= bits;                                     // 8 x iters evaluations of 8 different operand pairs: iters must be even
        }
    }
    if (acc != 0) atomicAdd(bad, 1u);
}

extern "C" int pk_victim_launch(int form, const void* in, void* bad, int blocks, int iters, hipStream_t s) {
    const v2f* p = static_cast<const v2f*>(in);
    unsigned* q = static_cast<unsigned*>(bad);
    switch (form) {
        case 0: hipLaunchKernelGGL(pk_victim<0>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 1: hipLaunchKernelGGL(pk_victim<1>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 2: hipLaunchKernelGGL(pk_victim<2>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 3: hipLaunchKernelGGL(pk_victim<3>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 4: hipLaunchKernelGGL(pk_victim<4>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 5: hipLaunchKernelGGL(pk_victim<5>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 6: hipLaunchKernelGGL(pk_victim<6>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 7: hipLaunchKernelGGL(pk_victim<7>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 8: hipLaunchKernelGGL(pk_victim<8>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 9: hipLaunchKernelGGL(pk_victim<9>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 10: hipLaunchKernelGGL(pk_victim<10>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 11: hipLaunchKernelGGL(pk_victim<11>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 12: hipLaunchKernelGGL(pk_victim<12>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 13: hipLaunchKernelGGL(pk_victim<13>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 14: hipLaunchKernelGGL(pk_victim<14>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 15: hipLaunchKernelGGL(pk_victim<15>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        case 16: hipLaunchKernelGGL(pk_victim<16>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
        default: hipLaunchKernelGGL(pk_victim<17>, dim3(blocks), dim3(256), 0, s, p, q, iters); break;
    }
    return (int)hipGetLastError();
}
