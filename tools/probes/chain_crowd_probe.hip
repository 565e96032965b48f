// Probe (MI355X): the 16-pivot chain on wave 0 of a 512-thread workgroup while the waves on the OTHER three SIMDs (1, 2, 3, 5, 6, 7:
// not wave 4, its SIMD partner) are busy with what the factorisation's other rows do beside a chain.
//   hipcc --offload-arch=gfx950 -O3 -I aruco_slam_amd/csrc tools/probes/chain_crowd_probe.hip -o /tmp/chain_crowd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ekf_solve_device.h"
// kind: 0 nothing, 1 f64 MFMA loop, 2 LDS block reads, 3 LDS flag poll + s_sleep 1, 4 LDS get + 4 MFMA + LDS put (a trailing update),
// 5 coherent global load poll + s_sleep 4 (exchange polling), 6 f64 FMA loop, 7 s_sleep loop only
__global__ __launch_bounds__(512) void k(double* io, long long* out, int kind, unsigned mask, int reps) {
    __shared__ volatile int stop;
    __shared__ volatile int flag;
    __shared__ double blk[8][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    if (threadIdx.x == 0) { stop = 0; flag = 0; }
    for (int e = threadIdx.x; e < 8 * 256; e += 512) blk[e >> 8][e & 255] = 1.0 / (1 + e);
    __syncthreads();
    if (wave == 0) {
        sf64x4 m0;
        for (int r = 0; r < 4; ++r) m0[r] = io[(g + 4 * r) * 16 + c];
        sf64x4 acc = {0, 0, 0, 0};
        const long long t0 = clock64();
        for (int it = 0; it < reps; ++it) {
            SvChain s; s.m = m0; sf64x4 xop;
            s.bad = sv_chain_t<false>(s, xop, c, g);
            for (int r = 0; r < 4; ++r) acc[r] += xop[r];
        }
        const long long t1 = clock64();
        if (lane == 0) { out[0] = t1 - t0; stop = 1; }
        for (int r = 0; r < 4; ++r) io[256 + (g + 4 * r) * 16 + c] = acc[r];
    } else if ((mask >> wave) & 1u) {
        sf64x4 a = {1.0, 2.0, 3.0, 4.0};
        double x = 1.0 + lane;
        volatile ekf_lds_int* lf = ekf_lds_flags((void*)&flag);
        while (!stop) {
            if (kind == 1) for (int j = 0; j < 4; ++j) a = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a, 0, 0, 0);
            if (kind == 2) { const sf64x4 v = sv_lds_get(blk[wave], lane); a += v; }
            if (kind == 3) { if (lf[0] == 0) __builtin_amdgcn_s_sleep(1); }
            if (kind == 4) { const sf64x4 v = sv_lds_get(blk[wave], lane); sv_mm_sub(a, v, v); sv_lds_put(blk[wave], a, lane); }
            if (kind == 5) { x += ekf_ldc(io + 1024 + lane); __builtin_amdgcn_s_sleep(4); }
            if (kind == 6) for (int j = 0; j < 16; ++j) x = __builtin_fma(x, 1.0000001, 1e-9);
            if (kind == 7) __builtin_amdgcn_s_sleep(1);
        }
        io[2048 + threadIdx.x] = a[0] + x;
    }
}
int main() {
    double h[512] = {0};
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
    double* d; long long* o; hipMalloc(&d, 4096 * 8); hipMalloc(&o, 64);
    hipMemset(d, 0, 4096 * 8);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const char* kinds[8] = {"idle", "f64 MFMA loop", "LDS block reads", "LDS flag poll + s_sleep 1", "LDS get + 4 MFMA + LDS put", "coherent global poll + s_sleep 4", "f64 FMA loop", "s_sleep 1 loop"};
    const unsigned masks[3] = {0xEEu, 0x10u, 0xFEu};      // other SIMDs only / the SIMD partner only / everybody
    const char* mnames[3] = {"waves 1 2 3 5 6 7 (other SIMDs)", "wave 4 (same SIMD)", "waves 1..7"};
    for (int mi = 0; mi < 3; ++mi)
        for (int kind = 0; kind < 8; ++kind) {
            long long r = 0;
            for (int trial = 0; trial < 2; ++trial) {
                hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, o, kind, masks[mi], 100);
                hipDeviceSynchronize();
                hipMemcpy(&r, o, 8, hipMemcpyDeviceToHost);
            }
            printf("%-34s %-34s: chain on wave 0 %7.1f cycles\n", mnames[mi], kinds[kind], r / 100.0);
        }
    return 0;
}
