// Is v_mfma_f32_16x16x4_f32 a chain of four fused multiply-adds over k ascending (like two v_mfma_f32_32x32x2_f32 over the same
// k's, which the covariance update uses)?  Compares the instruction with fmaf chains on random data of mixed magnitudes.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_order_probe tools/probes/mfma_order_probe.hip && /tmp/mfma_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A [16][K], B [K][16], C [16][16] row-major; K multiple of 4.  out16: 16x16x4 over k ascending; out32: the same through
// 32x32x2 (the 16x16 problem embedded in the top-left corner of a 32x32 tile)
__global__ void probe(const float* A, const float* B, const float* C, int K, float* out16, float* out32, float* outf) {
    const int l = threadIdx.x, j = l & 15, g = l >> 4;
    f4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(4 * g + r) * 16 + j];
    for (int k0 = 0; k0 < K; k0 += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j * K + k0 + g], B[(k0 + g) * 16 + j], acc, 0, 0, 0);   // A: lane (i = l & 15, k = l >> 4)
    for (int r = 0; r < 4; ++r) out16[(4 * g + r) * 16 + j] = acc[r];
    const int l31 = l & 31, lhi = l >> 5;
    f16v a2;
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * lhi;
        a2[reg] = (row < 16 && l31 < 16) ? C[row * 16 + l31] : 0.f;
    }
    for (int k0 = 0; k0 < K; k0 += 2) {
        const float av = l31 < 16 ? A[l31 * K + k0 + lhi] : 0.f, bv = l31 < 16 ? B[(k0 + lhi) * 16 + l31] : 0.f;
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, a2, 0, 0, 0);
    }
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * lhi;
        if (row < 16 && l31 < 16) out32[row * 16 + l31] = a2[reg];
    }
    for (int e = l; e < 256; e += 64) {
        const int i = e >> 4, jj = e & 15;
        float c = C[e];
        for (int k = 0; k < K; ++k) c = __builtin_fmaf(A[i * K + k], B[k * 16 + jj], c);
        outf[e] = c;
    }
}
int main() {
    const int K = 96, trials = 2000;
    std::vector<float> A(16 * K), B(K * 16), C(256), o16(256), o32(256), of(256);
    float *dA, *dB, *dC, *d16, *d32, *df;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&d16, 1024); hipMalloc(&d32, 1024); hipMalloc(&df, 1024);
    srand(1);
    auto rnd = [&](int spread) { float v = (float)rand() / RAND_MAX * 2.f - 1.f; int e = rand() % (2 * spread + 1) - spread; return ldexpf(v, e); };
    long bad16 = 0, bad32 = 0, bad1632 = 0;
    for (int t = 0; t < trials; ++t) {
        const int spread = t % 3 == 0 ? 0 : (t % 3 == 1 ? 6 : 20);
        for (auto& v : A) v = rnd(spread);
        for (auto& v : B) v = rnd(spread);
        for (auto& v : C) v = rnd(spread);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, d16, d32, df);
        hipMemcpy(o16.data(), d16, 1024, hipMemcpyDeviceToHost); hipMemcpy(o32.data(), d32, 1024, hipMemcpyDeviceToHost); hipMemcpy(of.data(), df, 1024, hipMemcpyDeviceToHost);
        for (int e = 0; e < 256; ++e) {
            bad16 += memcmp(&o16[e], &of[e], 4) != 0;
            bad32 += memcmp(&o32[e], &of[e], 4) != 0;
            bad1632 += memcmp(&o16[e], &o32[e], 4) != 0;
        }
    }
    printf("elements compared: %d\n16x16x4 != fmaf chain: %ld\n32x32x2 != fmaf chain: %ld\n16x16x4 != 32x32x2: %ld\n", trials * 256, bad16, bad32, bad1632);
    return 0;
}
