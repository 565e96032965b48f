// Diagnostics (MI355X): does a write-through (sc1) store leave a DIRTY line behind in its XCD's L2 that is
// written back again later, after another XCD has overwritten the location?  Kernels back to back, no host sync.
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_writeback_probe.hip -o aruco_slam_amd/build/xcd_wb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SC1>
__global__ void writer(double* data, int n, double v, int xcd) {
    if ((int)(blockIdx.x & 7) != xcd) return;
    for (int e = threadIdx.x + (blockIdx.x >> 3) * blockDim.x; e < n; e += blockDim.x * (gridDim.x >> 3)) {
        if (SC1) __hip_atomic_store(data + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else data[e] = v;
    }
}
__global__ void thrash(const float* big, size_t n, float* sink) {       // stream through every L2
    float acc = 0.f;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) acc += big[e];
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void reader(const double* data, int n, double expect, int* bad) {
    int nb = 0;
    for (int e = threadIdx.x; e < n; e += blockDim.x)
        nb += (__hip_atomic_load(data + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != expect);
    if (nb) atomicAdd(bad + (blockIdx.x & 7), nb);
}
int main() {
    const int n = 32768;      // 256 KB
    const size_t nbig = (size_t)256 << 20;    // 1 GB of floats / 4
    double* data; int* bad; float* big; float* sink;
    hipMalloc(&data, n * 8); hipMalloc(&bad, 64); hipMalloc(&big, nbig * 4); hipMalloc(&sink, 64);
    hipMemset(big, 0, nbig * 4);
    hipMemset(data, 0, n * 8);
    hipDeviceSynchronize();
    for (int first_sc1 = 0; first_sc1 < 2; ++first_sc1) {
        int total = 0, hb[8];
        for (int trial = 0; trial < 30; ++trial) {
            hipMemsetAsync(bad, 0, 64, 0);
            const double A = 100.0 + trial, B = 5000.0 + trial;
            if (first_sc1) hipLaunchKernelGGL(writer<1>, dim3(64), dim3(256), 0, 0, data, n, A, 0);
            else hipLaunchKernelGGL(writer<0>, dim3(64), dim3(256), 0, 0, data, n, A, 0);
            hipLaunchKernelGGL(writer<1>, dim3(64), dim3(256), 0, 0, data, n, B, 3);      // another XCD overwrites
            hipLaunchKernelGGL(thrash, dim3(2048), dim3(256), 0, 0, big, nbig, sink);     // evict everything
            hipLaunchKernelGGL(reader, dim3(8), dim3(256), 0, 0, data, n, B, bad);
            hipDeviceSynchronize();
            hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
            for (int i = 0; i < 8; ++i) total += hb[i];
        }
        printf("first writer %s (XCD 0), second writer sc1 (XCD 3), L2 thrashed, coherent read: words that reverted to the FIRST value over 30 trials: %d\n",
               first_sc1 ? "sc1" : "plain", total);
    }
    return 0;
}
