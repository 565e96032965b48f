// Micro-benchmark (diagnostics, MI355X): how fast does data written by one workgroup become
// visible to workgroups on other XCDs inside ONE launch, per store / load flavour?
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_exchange_probe.hip -o aruco_slam_amd/build/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define SENT 0xFFFBADC0FFFBADC0ull
__device__ __forceinline__ bool is_sent(double v) { return __double_as_longlong(v) == (long long)SENT; }
template <int MODE> __device__ __forceinline__ void st(double* p, double v) {
    if (MODE == 0 || MODE == 3) *p = v;
    else if (MODE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __builtin_nontemporal_store(v, p);
}
template <int MODE> __device__ __forceinline__ double ld(const double* p) {
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int SM, int LM>
__global__ __launch_bounds__(512) void probe(double* data, unsigned long long* flag, long long* out, int ndbl) {
    const int tid = threadIdx.x, w = blockIdx.x;
    if (w == 0) {
        const long long t0 = wall_clock64();
        for (int e = tid; e < ndbl; e += 512) st<SM>(data + e, 1.0 + e);
        if (SM == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const long long t1 = wall_clock64();
            __hip_atomic_store(flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[0] = t0; out[1] = t1;
        }
        return;
    }
    if (tid >= 64) return;
    int it = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull && ++it < 2000000) __builtin_amdgcn_s_sleep(4);
    const long long t1 = wall_clock64();
    int nsent = 0;
    double acc = 0.0;
    for (int base = 0; base < ndbl; base += 64 * 8) {      // 8 loads in flight
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (base + 64 * u + tid < ndbl) ? ld<LM>(data + base + 64 * u + tid) : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) { nsent += is_sent(v[u]); acc += v[u]; }
    }
    const long long t2 = wall_clock64();
    for (int o = 32; o > 0; o >>= 1) nsent += __shfl_xor(nsent, o);
    if (tid == 0) { out[4 * w] = t1; out[4 * w + 1] = t2; out[4 * w + 2] = nsent; out[4 * w + 3] = (long long)acc; }
}
template <int SM, int LM> void run(const char* name, double* data, unsigned long long* flag, long long* out, int ndbl, int nwg) {
    std::vector<long long> h(4 * nwg);
    for (int trial = 0; trial < 3; ++trial) {
        hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(data), (int)0xFFFBADC0u, (size_t)ndbl * 2);
        hipMemset(flag, 0, 8);
        hipMemset(out, 0, 4 * nwg * 8);
        hipDeviceSynchronize();
        hipLaunchKernelGGL((probe<SM, LM>), dim3(nwg), dim3(512), 0, 0, data, flag, out, ndbl);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, 4 * nwg * 8, hipMemcpyDeviceToHost);
        if (trial == 0) continue;
        double wr = (h[1] - h[0]) / 100.0, lag = 0, rd = 0; long long ns = 0;
        for (int w = 1; w < nwg; ++w) { lag += (h[4 * w] - h[1]) / 100.0; rd += (h[4 * w + 1] - h[4 * w]) / 100.0; ns += h[4 * w + 2]; }
        printf("%-34s %6.1f KB  write+drain %6.2f us  flag lag %5.2f us  read %6.2f us/consumer  stale words %lld (same-XCD wg8: %lld)\n",
               name, ndbl * 8 / 1024.0, wr, lag / (nwg - 1), rd / (nwg - 1), ns, h[4 * 8 + 2]);
    }
}
int main() {
    const int nwg = 17, maxd = 8192;
    double* data; unsigned long long* flag; long long* out;
    hipMalloc(&data, maxd * 8); hipMalloc(&flag, 256); hipMalloc(&out, 4 * nwg * 8);
    for (int ndbl : {256, 2048, 8192}) {
        run<0, 0>("plain store / cacheable load", data, flag, out, ndbl, nwg);
        run<1, 0>("sc1 store / cacheable load", data, flag, out, ndbl, nwg);
        run<2, 0>("sc0sc1 store / cacheable load", data, flag, out, ndbl, nwg);
        run<3, 0>("plain + wbl2 / cacheable load", data, flag, out, ndbl, nwg);
        run<4, 0>("nontemporal store / cacheable", data, flag, out, ndbl, nwg);
        run<0, 1>("plain store / sc1 load", data, flag, out, ndbl, nwg);
        run<1, 1>("sc1 store / sc1 load", data, flag, out, ndbl, nwg);
        run<2, 2>("sc0sc1 store / sc0sc1 load", data, flag, out, ndbl, nwg);
        run<3, 1>("plain + wbl2 / sc1 load", data, flag, out, ndbl, nwg);
    }
    return 0;
}
