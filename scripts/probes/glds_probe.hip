// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950 — lane -> LDS slot mapping and out-of-range behaviour.
// hipcc --offload-arch=gfx950 -O3 scripts/probes/glds_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const char* src, float* out, int nbytes) {
    __shared__ __attribute__((aligned(16))) char buf[2048];
    for (int i = threadIdx.x; i < 512; i += 64) reinterpret_cast<float*>(buf)[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    int voff = (63 - threadIdx.x) * 16;          // lane l reads source slot 63 - l
    if (threadIdx.x % 3 == 1) voff = 0x7ffffff0;  // out of range -> should deposit zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(buf + 1024), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = reinterpret_cast<float*>(buf)[i];
}
int main() {
    float h[256], *d, *o, ho[512];
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>((const char*)d, o, (int)sizeof(h));
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (ho[i] != -7.f) ++bad;                 // first KiB untouched
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const float want = (l % 3 == 1) ? 0.f : (float)((63 - l) * 4 + j);
            if (ho[256 + l * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got %g want %g\n", l, j, ho[256 + l * 4 + j], want); ++bad; }
        }
    printf("glds probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK: LDS slot = base + lane*16, out-of-range lanes write zeros", bad);
    return bad != 0;
}
