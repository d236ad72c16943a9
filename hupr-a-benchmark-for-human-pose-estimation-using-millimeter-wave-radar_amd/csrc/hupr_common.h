// Shared host/device helpers for the gfx950 kernels behind include/hupr.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "hupr.h"
#include "hupr_debug.h"

namespace hupr {

// thread-local error text returned by hupr_last_error()
char* error_buffer();
inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HUPR_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return ::hupr::fail(HUPR_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define HUPR_LAUNCH_OK(name)                                                            \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess)                                                          \
            return ::hupr::fail(HUPR_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// Every kernel launch of the library goes through HUPR_LAUNCH: a process-wide relaxed counter (hupr_launch_count) lets bench.py
// print the step's launch count without a tracer ("730 launches per step").
unsigned long long* launch_counter();
#define HUPR_LAUNCH(...)                                              \
    do {                                                              \
        __atomic_fetch_add(::hupr::launch_counter(), 1ull, __ATOMIC_RELAXED); \
        hipLaunchKernelGGL(__VA_ARGS__);                              \
    } while (0)

// ---- device helpers -------------------------------------------------------------------
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// acc += a*b
__device__ __forceinline__ void cfma(float2& acc, float2 a, float2 b) {
    acc.x = fmaf(a.x, b.x, fmaf(-a.y, b.y, acc.x));
    acc.y = fmaf(a.x, b.y, fmaf(a.y, b.x, acc.y));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Activation storage helpers: 4 consecutive channels as fp32 (16 B) or bf16 (8 B), always computed on as fp32.
typedef __bf16 hupr_bf16x4 __attribute__((ext_vector_type(4)));
typedef float hupr_f32x4 __attribute__((ext_vector_type(4)));
// HUPR_NT_ACT_LOADS (A/B build): the element-wise / statistics kernels read their activations with the non-temporal hint
#ifdef HUPR_NT_ACT_LOADS
#define HUPR_ACT_LD(TYPE_, PTR_) __builtin_nontemporal_load(reinterpret_cast<const TYPE_*>(PTR_))
#else
#define HUPR_ACT_LD(TYPE_, PTR_) (*reinterpret_cast<const TYPE_*>(PTR_))
#endif
template <typename T> __device__ __forceinline__ float4 ld_act4(const T* p);
template <> __device__ __forceinline__ float4 ld_act4<float>(const float* p) {
    const hupr_f32x4 t = HUPR_ACT_LD(hupr_f32x4, p);
    return make_float4(t[0], t[1], t[2], t[3]);
}
template <> __device__ __forceinline__ float4 ld_act4<__bf16>(const __bf16* p) {
    const hupr_bf16x4 v = HUPR_ACT_LD(hupr_bf16x4, p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st_act4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st_act4(__bf16* p, float4 v) {
    const hupr_bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<hupr_bf16x4*>(p) = o;
}

// 16-byte activation vectors: V consecutive channels (4 fp32 / 8 bf16) loaded to / stored from fp32 registers
template <typename T> struct ActVec;
template <> struct ActVec<float> {
    static constexpr int V = 4;
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const hupr_f32x4 t = HUPR_ACT_LD(hupr_f32x4, p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct ActVec<__bf16> {
    static constexpr int V = 8;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ void load(const __bf16* p, float* v) {
        const bf16x8 t = HUPR_ACT_LD(bf16x8, p);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)t[k];
    }
    static __device__ __forceinline__ void store(__bf16* p, const float* v) {
        bf16x8 t;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (__bf16)v[k];
        *reinterpret_cast<bf16x8*>(p) = t;
    }
};

static inline hipStream_t as_stream(hupr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace hupr
