// Diagnostics (MI355X): can a cacheable load in kernel C see a STALE L2 copy of a line that its
// XCD cached in kernel A and another XCD overwrote (write-through, sc1) in kernel B?
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_stale_probe.hip -o aruco_slam_amd/build/xcd_stale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE> __device__ __forceinline__ double ld(const double* p) {
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__global__ void reader(const double* data, int n, double expect, int* bad) {      // one WG per XCD (grid 8)
    int nb = 0;
    for (int e = threadIdx.x; e < n; e += blockDim.x) nb += (ld<MODE>(data + e) != expect);
    if (nb) atomicAdd(bad + blockIdx.x, nb);
}
template <int SM>
__global__ void writer(double* data, int n, double v, int writer_wg) {
    if ((int)blockIdx.x != writer_wg) return;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        if (SM == 0) data[e] = v;
        else __hip_atomic_store(data + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main() {
    const int n = 4096;       // 32 KB
    double* data; int* bad;
    hipMalloc(&data, n * 8); hipMalloc(&bad, 64);
    std::vector<double> h(n, 1.0);
    for (int sm = 0; sm < 2; ++sm)
        for (int lm = 0; lm < 2; ++lm) {
            int stale_total = 0, hb[8];
            for (int trial = 0; trial < 20; ++trial) {
                const double v1 = 10.0 + trial, v2 = 1000.0 + trial;
                for (auto& x : h) x = v1;
                hipMemcpy(data, h.data(), n * 8, hipMemcpyHostToDevice);
                hipMemset(bad, 0, 64);
                // A: every XCD caches the buffer (cacheable loads)
                hipLaunchKernelGGL(reader<0>, dim3(8), dim3(256), 0, 0, data, n, v1, bad);
                // B: one XCD overwrites it
                if (sm == 0) hipLaunchKernelGGL(writer<0>, dim3(8), dim3(256), 0, 0, data, n, v2, 3);
                else hipLaunchKernelGGL(writer<1>, dim3(8), dim3(256), 0, 0, data, n, v2, 3);
                hipMemsetAsync(bad, 0, 64, 0);
                // C: every XCD reads again
                if (lm == 0) hipLaunchKernelGGL(reader<0>, dim3(8), dim3(256), 0, 0, data, n, v2, bad);
                else hipLaunchKernelGGL(reader<1>, dim3(8), dim3(256), 0, 0, data, n, v2, bad);
                hipDeviceSynchronize();
                hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
                for (int i = 0; i < 8; ++i) stale_total += hb[i];
            }
            printf("writer %s store, reader %s load: stale words over 20 trials (8 XCDs x 4096): %d\n",
                   sm ? "sc1" : "plain", lm ? "sc1" : "cacheable", stale_total);
        }
    return 0;
}
