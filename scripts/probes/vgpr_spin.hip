// Probe: a do-nothing aggressor that only OCCUPIES registers.  spin_<N> claims N VGPRs per lane (clobber of v<N-1>) and loops on a few
// VALU adds for `iters` rounds; scripts/interp_race.py SPIN=1 runs the SLP-built resampling kernel beside each of them to see whether the
// corruption depends on what the co-resident wave executes or only on where the victim's registers land in the SIMD's register file.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void spin_56(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v55"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_64(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v63"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_72(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v71"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_80(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v79"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_88(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v87"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_96(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v95"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_104(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v103"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_112(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v111"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_120(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v119"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_128(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v127"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_160(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v159"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_192(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v191"); }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_224(int iters, float* sink) {
    float a = threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "v223"); }
    if (a == -1.0f) sink[0] = a;
}

// The same occupancy (112 VGPRs, 47 KB of LDS, 256 threads) with ONE instruction class in the loop: what does the co-resident wave have to
// execute for the victim's packed sequence to go wrong?
typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
typedef float f32x16_ __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(256) void spin_mfma(int iters, float* sink) {
    bf16x8_ a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)0.5f; }
    f32x16_ c0 = {0}, c1 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        asm volatile("" ::: "v111");
    }
    if (c0[0] + c1[3] == -1.0f) sink[0] = c0[0];
}
extern "C" __global__ __launch_bounds__(256) void spin_lds(int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned buf[47 * 256];
    for (int i = threadIdx.x; i < 47 * 256; i += 256) buf[i] = i;
    __syncthreads();
    u32x4_ acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        const u32x4_ v = *reinterpret_cast<const u32x4_*>(&buf[((threadIdx.x * 4 + i * 64) % (47 * 256 - 4)) & ~3]);
        acc += v;
        asm volatile("" ::: "v111");
    }
    if (acc[0] == 0xffffffffu) sink[0] = 1.f;
}
extern "C" __global__ __launch_bounds__(256) void spin_ldsw(int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned buf[47 * 256];
    u32x4_ v = {threadIdx.x, 1, 2, 3};
    for (int i = 0; i < iters; ++i) {
        *reinterpret_cast<u32x4_*>(&buf[((threadIdx.x * 4 + i * 64) % (47 * 256 - 4)) & ~3]) = v;
        v[1] += 1;
        asm volatile("" ::: "v111");
    }
    __syncthreads();
    if (buf[threadIdx.x] == 0xffffffffu) sink[0] = 1.f;
}
extern "C" __global__ __launch_bounds__(256) void spin_barrier(int iters, float* sink) {
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        __syncthreads();
        asm volatile("v_add_f32 %0, %0, %0" : "+v"(a) : : "v111");
    }
    if (a == -1.0f) sink[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void spin_gload(int iters, float* sink, const unsigned* src, int n) {
    u32x4_ acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        acc += *reinterpret_cast<const u32x4_*>(src + ((((long)blockIdx.x * 256 + threadIdx.x) * 4 + (long)i * 65536) % (n - 4) & ~3L));
        asm volatile("" ::: "v111");
    }
    if (acc[0] == 0xffffffffu) sink[0] = 1.f;
}
extern "C" __global__ __launch_bounds__(256) void spin_cvt(int iters, float* sink) {
    float a = threadIdx.x, b = 0.25f;
    unsigned r = 0;
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_permlane32_swap_b32 %0, %0" : "+v"(r) : "v"(a), "v"(b) : "v111");
        a += 1.0f;
    }
    if (r == 0xffffffffu) sink[0] = 1.f;
}
extern "C" int spin_class_launch(int which, int iters, int blocks, void* sink, const void* src, int n, void* stream) {
    hipStream_t s = (hipStream_t)stream; float* k = (float*)sink;
    switch (which) {
        case 0: hipLaunchKernelGGL(spin_mfma, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 1: hipLaunchKernelGGL(spin_lds, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 2: hipLaunchKernelGGL(spin_ldsw, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 3: hipLaunchKernelGGL(spin_barrier, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 4: hipLaunchKernelGGL(spin_gload, dim3(blocks), dim3(256), 0, s, iters, k, (const unsigned*)src, n); break;
        case 5: hipLaunchKernelGGL(spin_cvt, dim3(blocks), dim3(256), 0, s, iters, k); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
extern "C" int spin_launch(int nv, int iters, int blocks, void* sink, void* stream) {
    hipStream_t s = (hipStream_t)stream; float* k = (float*)sink;
    switch (nv) {
        case 56: hipLaunchKernelGGL(spin_56, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 64: hipLaunchKernelGGL(spin_64, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 72: hipLaunchKernelGGL(spin_72, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 80: hipLaunchKernelGGL(spin_80, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 88: hipLaunchKernelGGL(spin_88, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 96: hipLaunchKernelGGL(spin_96, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 104: hipLaunchKernelGGL(spin_104, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 112: hipLaunchKernelGGL(spin_112, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 120: hipLaunchKernelGGL(spin_120, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 128: hipLaunchKernelGGL(spin_128, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 160: hipLaunchKernelGGL(spin_160, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 192: hipLaunchKernelGGL(spin_192, dim3(blocks), dim3(256), 0, s, iters, k); break;
        case 224: hipLaunchKernelGGL(spin_224, dim3(blocks), dim3(256), 0, s, iters, k); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
