// Probe (MI355X): which waves of a 512-thread workgroup share a SIMD with wave 0, and what a co-resident
// wave's f64 MFMA / f64 FMA / LDS-poll loop costs the 16-pivot chain running on wave 0.
//   hipcc --offload-arch=gfx950 -O3 -I aruco_slam_amd/csrc tools/probes/simd_share_probe.hip -o aruco_slam_amd/build/simd_share_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ekf_solve_device.h"
// other: wave that runs the disturbing loop; kind: 0 nothing, 1 f64 MFMA, 2 f64 FMA, 3 LDS poll with s_sleep 1
__global__ __launch_bounds__(512) void k(double* io, long long* out, int other, int kind, int reps) {
    __shared__ volatile int stop;
    __shared__ volatile int flag;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    if (threadIdx.x == 0) { stop = 0; flag = 0; }
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
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (lane == 0) out[1] = hw;
    } else if (wave == other) {
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (lane == 0) out[2] = hw;
        sf64x4 a = {1.0, 2.0, 3.0, 4.0};
        double x = 1.0 + lane;
        long long n = 0;
        while (!stop) {
            if (kind == 1) for (int j = 0; j < 4; ++j) a = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a, 0, 0, 0);
            if (kind == 2) for (int j = 0; j < 16; ++j) x = __builtin_fma(x, 1.0000001, 1e-9);
            if (kind == 3) { while (flag == 0 && !stop) __builtin_amdgcn_s_sleep(1); }
            ++n;
        }
        io[512 + lane] = a[0] + x;
        if (lane == 0) out[3] = n;
    }
}
int main() {
    double h[512] = {0};
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
    double* d; long long* o; hipMalloc(&d, 1024 * 8); hipMalloc(&o, 64);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const char* kinds[4] = {"idle", "f64 MFMA loop", "f64 FMA loop", "LDS poll + s_sleep 1"};
    for (int kind = 0; kind < 4; ++kind)
        for (int other : {1, 4}) {
            long long r[4] = {0, 0, 0, 0};
            for (int trial = 0; trial < 2; ++trial) {
                hipMemset(o, 0, 64);
                hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, o, other, kind, 100);
                hipDeviceSynchronize();
                hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
            }
            printf("wave %d %-22s: chain on wave 0 %7.1f cycles   (HW_ID simd: wave 0 = %lld, wave %d = %lld; loop iterations %lld)\n",
                   other, kinds[kind], r[0] / 100.0, (r[1] >> 4) & 3, other, (r[2] >> 4) & 3, r[3]);
        }
    return 0;
}
