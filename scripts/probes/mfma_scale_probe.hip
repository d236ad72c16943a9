// Operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3), determined by experiment
// (csrc/attention_mx8.hip relies on them): hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_scale_probe scripts/probes/mfma_scale_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OA, int OB>
__global__ void k_mfma(const unsigned char* a, const unsigned char* b, const int* sa, const int* sb, float* d) {
    const int l = threadIdx.x;
    i32x8 av, bv;
    memcpy(&av, a + l * 32, 32);
    memcpy(&bv, b + l * 32, 32);
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, OA, sa[l], OB, sb[l]);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}

static std::vector<float> run(int oa, int ob, const std::vector<unsigned char>& A, const std::vector<unsigned char>& B,
                              const std::vector<int>& SA, const std::vector<int>& SB) {
    unsigned char *da, *db; int *dsa, *dsb; float* dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 4096);
    hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, SB.data(), 256, hipMemcpyHostToDevice);
    if (oa == 0 && ob == 0) k_mfma<0, 0><<<1, 64>>>(da, db, dsa, dsb, dd);
    else if (oa == 1 && ob == 0) k_mfma<1, 0><<<1, 64>>>(da, db, dsa, dsb, dd);
    else if (oa == 2 && ob == 0) k_mfma<2, 0><<<1, 64>>>(da, db, dsa, dsb, dd);
    else if (oa == 3 && ob == 0) k_mfma<3, 0><<<1, 64>>>(da, db, dsa, dsb, dd);
    else k_mfma<0, 1><<<1, 64>>>(da, db, dsa, dsb, dd);
    std::vector<float> D(1024);
    hipMemcpy(D.data(), dd, 4096, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dd);
    // D as [row][col] with the documented 32x32 C/D map: lane l, reg r -> row (r&3) + 8 (r>>2) + 4 (l>>5), col l & 31
    std::vector<float> M(1024);
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) M[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = D[l * 16 + r];
    return M;
}

int main() {
    const unsigned char ONE = 0x38;      // e4m3 1.0
    std::vector<int> S127(64, 127);
    // 1. which (lane, byte) of B meets (lane 0 / 32, byte 0 / 5 / 17) of A?
    for (int la : {0, 32, 3, 35})
        for (int ma : {0, 5, 17}) {
            std::vector<unsigned char> A(2048, 0);
            A[la * 32 + ma] = ONE;
            printf("A one-hot (lane %2d, byte %2d): ", la, ma);
            int hits = 0;
            for (int lb : {0, 32})
                for (int mb = 0; mb < 32; ++mb) {
                    std::vector<unsigned char> B(2048, 0);
                    B[lb * 32 + mb] = ONE;
                    auto M = run(0, 0, A, B, S127, S127);
                    for (int i = 0; i < 1024; ++i)
                        if (M[i] != 0.f) { printf("meets B (lane %2d, byte %2d) -> D[%d][%d] = %g; ", lb, mb, i / 32, i % 32, M[i]); ++hits; }
                }
            printf("%s\n", hits ? "" : "no match");
        }
    // 2. all ones: D = 64 everywhere?
    std::vector<unsigned char> A1(2048, ONE), B1(2048, ONE);
    {
        auto M = run(0, 0, A1, B1, S127, S127);
        printf("all ones, scales 127: D[0][0] = %g, D[31][31] = %g, D[5][7] = %g\n", M[0], M[1023], M[5 * 32 + 7]);
        std::vector<int> S0(64, 0);
        M = run(0, 0, A1, B1, S0, S0);
        printf("all ones, scale VGPRs 0 (run-time zero): D[0][0] = %g\n", M[0]);
    }
    // 3. one lane's A scale = 128 (x2) in byte 0: which entries change?
    for (int ls : {0, 5, 32, 37}) {
        std::vector<int> SA(64, 127);
        SA[ls] = 128;
        auto M = run(0, 0, A1, B1, SA, S127);
        printf("scale_a byte 0 of lane %2d = 128: rows with D != 64:", ls);
        for (int i = 0; i < 32; ++i)
            if (M[i * 32] != 64.f) printf(" row %d -> %g", i, M[i * 32]);
        printf("  (cols uniform: %s)\n", M[ls % 32 * 32] == M[ls % 32 * 32 + 17] ? "yes" : "no");
    }
    for (int ls : {0, 5, 32, 37}) {
        std::vector<int> SB(64, 127);
        SB[ls] = 128;
        auto M = run(0, 0, A1, B1, S127, SB);
        printf("scale_b byte 0 of lane %2d = 128: cols with D != 64:", ls);
        for (int j = 0; j < 32; ++j)
            if (M[j] != 64.f) printf(" col %d -> %g", j, M[j]);
        printf("\n");
    }
    // 4. op_sel: scale byte in byte position p of the VGPR, selected by opsel = p?
    for (int p = 1; p < 4; ++p) {
        std::vector<int> SA(64, 127 | (127 << 8) | (127 << 16) | (127 << 24));
        SA[5] = (SA[5] & ~(0xff << (8 * p))) | (129 << (8 * p));
        auto M = run(p, 0, A1, B1, SA, S127);
        printf("opsel_a = %d, byte %d of lane 5 = 129 (x4): D[5][0] = %g, D[6][0] = %g\n", p, p, M[5 * 32], M[6 * 32]);
    }
    // 5. within a lane's 32 bytes: does the half (bytes 0..15 / 16..31) matter for the scale?  A lane 5: bytes 0..15 = 1, rest 0
    {
        std::vector<unsigned char> A(2048, 0);
        for (int m = 0; m < 16; ++m) A[5 * 32 + m] = ONE;
        auto M = run(0, 0, A, B1, S127, S127);
        printf("A lane 5 bytes 0..15 ones: D[5][0] = %g\n", M[5 * 32]);
    }
    // 6. whose data does a lane's scale byte multiply?  A: only lane LA holds ones (its 32 bytes); scale 128 on lane LS
    for (int la : {5, 37})
        for (int ls : {5, 37}) {
            std::vector<unsigned char> A(2048, 0);
            for (int m = 0; m < 32; ++m) A[la * 32 + m] = ONE;
            std::vector<int> SA(64, 127);
            SA[ls] = 128;
            auto M = run(0, 0, A, B1, SA, S127);
            printf("A ones in lane %2d only, scale_a 128 on lane %2d: D[5][0] = %g (32 = not scaled, 64 = scaled)\n", la, ls, M[5 * 32]);
        }
    for (int lb : {5, 37})
        for (int ls : {5, 37}) {
            std::vector<unsigned char> B(2048, 0);
            for (int m = 0; m < 32; ++m) B[lb * 32 + m] = ONE;
            std::vector<int> SB(64, 127);
            SB[ls] = 128;
            auto M = run(0, 0, A1, B, S127, SB);
            printf("B ones in lane %2d only, scale_b 128 on lane %2d: D[0][5] = %g (32 = not scaled, 64 = scaled)\n", lb, ls, M[5]);
        }
    return 0;
}
