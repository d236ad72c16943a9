// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value == element index; every lane passes
// its own address; print what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(int mode, unsigned short* out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int elem;  // element index (u16 units) this lane's address points at
    if (mode == 0) elem = l * 4;                          // consecutive 8-byte segments
    else if (mode == 1) elem = (l & 15) * 64 + (l >> 4) * 4;   // lane -> row (l&15) of a [16][64] matrix, column group l>>4
    else elem = (l & 15) * 64;                            // all four 16-lane groups same rows
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(&lds[elem]));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 2) ? "\n" : "   |  ");
    }
    return 0;
}
