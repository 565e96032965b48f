// Probe (MI355X): does a kernel launched with hipExtAnyOrderLaunch (AQL barrier bit cleared) start while the
// previous kernel of the SAME stream is still running?  Two single-workgroup kernels that spin ~200 us each.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/anyorder_probe.hip -o aruco_slam_amd/build/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long* out, int slot, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * slot] = t0; out[2 * slot + 1] = wall_clock64(); }
}
int main() {
    long long* d; hipMalloc(&d, 64);
    hipStream_t s; hipStreamCreate(&s);
    for (int flags = 0; flags < 2; ++flags) {
        for (int trial = 0; trial < 3; ++trial) {
            hipMemsetAsync(d, 0, 64, s);
            hipStreamSynchronize(s);
            auto w0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 0, 20000LL);                            // 200 us at 100 MHz
            hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, d, 1, 20000LL);
            hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
            long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
            if (trial) printf("flags=%d  wall %.0f us  kernel B started %.1f us after kernel A started (A ran %.1f us)\n", flags, us,
                              (h[2] - h[0]) / 100.0, (h[1] - h[0]) / 100.0);
        }
    }
    return 0;
}
