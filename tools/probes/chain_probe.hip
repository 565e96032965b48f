// Probe (MI355X): cycles of the 16-pivot chain (ekf_solve_device.h: sv_chain) on one wave, alone on its CU,
// and of its ingredients (s_memtime deltas; 100 repetitions each).
//   hipcc --offload-arch=gfx950 -O3 -I aruco_slam_amd/csrc tools/probes/chain_probe.hip -o aruco_slam_amd/build/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ekf_solve_device.h"

template <int MODE>
__global__ __launch_bounds__(512) void k(double* io, long long* out, int reps) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    if (threadIdx.x >= 64 * (MODE >= 10 ? 2 : 1)) return;         // MODE >= 10: a second wave runs the same thing
    sf64x4 m0;
    for (int r = 0; r < 4; ++r) m0[r] = io[(g + 4 * r) * 16 + c];
    sf64x4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < reps; ++it) {
        SvChain s;
        s.m = m0;
        sf64x4 xop = {0, 0, 0, 0};
        if ((MODE % 10) == 0) s.bad = sv_chain_t<false>(s, xop, c, g);
        if ((MODE % 10) == 1) {            // 16 x (readlane + rcp) only: the scalar path
            double d = 1.0;
            for (int j = 0; j < 16; ++j) { d = ekf_readlane_f64(s.m[j & 3], j) + d; d = sv_rcp(d); s.m[j & 3] += d; }
            xop = s.m;
        }
        if ((MODE % 10) == 2) {            // 16 x 8 fmac_dpp
            double nl = s.m[0];
            for (int j = 0; j < 16; ++j)
                asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %1, %1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %2, %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %3, %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %0, %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %1, %1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %2, %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %3, %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                             : "+v"(s.m[0]), "+v"(s.m[1]), "+v"(s.m[2]), "+v"(s.m[3]) : "v"(nl));
            xop = s.m;
        }
        if ((MODE % 10) == 3) {            // 16 group broadcasts
            double v = s.m[0];
            for (int j = 0; j < 16; ++j) v = sv_group_bcast<1>(v) + 1.0;
            xop[0] = v;
        }
        if ((MODE % 10) == 5) {            // 16 group broadcasts through ds_bpermute
            double v = s.m[0];
            const int addr = 4 * (16 + (lane & 15));
            for (int j = 0; j < 16; ++j) {
                const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
                v = __hiloint2double(hi, lo) + 1.0;
            }
            xop[0] = v;
        }
        if ((MODE % 10) == 6) {            // 16 x (ds_swizzle within 32 lanes + permlane32_swap)
            double v = s.m[0];
            for (int j = 0; j < 16; ++j) {
                int lo = __double2loint(v), hi = __double2hiint(v);
                lo = __builtin_amdgcn_ds_swizzle(lo, 0x000F | (16 << 5));      // bit mode: and 0x0F, or 0x10, xor 0
                hi = __builtin_amdgcn_ds_swizzle(hi, 0x000F | (16 << 5));
                auto q0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
                auto q1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
                v = __hiloint2double(q1[0], q0[0]) + 1.0;
            }
            xop[0] = v;
        }
        if ((MODE % 10) == 7) {            // 16 x readlane of 4 doubles (8 v_readlane_b32) + 4 fma with SGPR operands
            sf64x4 a = s.m;
            for (int j = 0; j < 16; ++j) {
                const double s0 = ekf_readlane_f64(a[0], j), s1 = ekf_readlane_f64(a[1], j + 16), s2 = ekf_readlane_f64(a[2], j + 32), s3 = ekf_readlane_f64(a[3], j + 48);
                a[0] = __builtin_fma(a[0], s0, a[1]); a[1] = __builtin_fma(a[1], s1, a[2]); a[2] = __builtin_fma(a[2], s2, a[3]); a[3] = __builtin_fma(a[3], s3, a[0]);
            }
            xop = a;
        }
        if ((MODE % 10) == 8) {            // 16 x LDS round trip (write 16 lanes, read all)
            __shared__ double buf[2][16];
            double v = s.m[0];
            for (int j = 0; j < 16; ++j) {
                if (g == 1) buf[j & 1][c] = v;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                v = buf[j & 1][c] + 1.0;
            }
            xop[0] = v;
        }
        if ((MODE % 10) == 4) {            // 16 x 8 plain v_fma_f64 (dependent in 8 chains)
            double nl = s.m[0];
            sf64x4 a = s.m, b = s.m;
            for (int j = 0; j < 16; ++j)
                for (int r = 0; r < 4; ++r) { a[r] = __builtin_fma(a[r], nl, a[r]); b[r] = __builtin_fma(b[r], nl, b[r]); }
            xop = a + b;
        }
        for (int r = 0; r < 4; ++r) acc[r] += xop[r];
    }
    const long long t1 = clock64();
    if (lane == 0 && threadIdx.x < 64) out[0] = t1 - t0;
    for (int r = 0; r < 4; ++r) io[256 + (g + 4 * r) * 16 + c] = acc[r];
}
int main() {
    double h[512] = {0};
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
    double* d; long long* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const int reps = 100;
    const char* names[9] = {"full chain (16 pivots)", "16 x (readlane + rcp), dependent", "16 x 8 v_fmac_f64_dpp", "16 x group broadcast (f64), dependent", "16 x 8 v_fma_f64",
                            "16 x group bcast via ds_bpermute", "16 x group bcast via ds_swizzle + permlane32", "16 x (4 readlane f64 + 4 fma)", "16 x group bcast via LDS write/read"};
    for (int two = 0; two < 1; ++two)
        for (int mode = 0; mode < 9; ++mode) {
            long long cyc = 0;
            for (int trial = 0; trial < 2; ++trial) {
                switch (mode + 10 * two) {
#define CASE(M) case M: hipLaunchKernelGGL(k<M>, dim3(1), dim3(512), 0, 0, d, o, reps); break;
                    CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
                }
                hipDeviceSynchronize();
                hipMemcpy(&cyc, o, 8, hipMemcpyDeviceToHost);
            }
            printf("%-40s %s: %8.1f cycles per repetition (%.1f per step)\n", names[mode], two ? "two waves " : "one wave  ", (double)cyc / reps, (double)cyc / reps / 16);
        }
    return 0;
}
