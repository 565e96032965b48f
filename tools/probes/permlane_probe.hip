// Probe (MI355X): semantics of the gfx950 cross-row swaps as the builtins expose them, and the
// "value of 16-lane group G0 in every group" broadcast built from them (ekf_solve_device.h: sv_group_bcast).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int G0> __device__ int group_bcast(int v) {
    auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const int x = (G0 & 1) ? p[1] : p[0];
    auto q = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (G0 & 2) ? q[1] : q[0];
}
__global__ void k(int* out) {
    const int lane = threadIdx.x, v = 1000 * (lane >> 4) + (lane & 15);      // group * 1000 + lane in group
    auto p = __builtin_amdgcn_permlane16_swap(v, v + 100000, false, false);
    auto q = __builtin_amdgcn_permlane32_swap(v, v + 100000, false, false);
    out[lane] = p[0]; out[64 + lane] = p[1]; out[128 + lane] = q[0]; out[192 + lane] = q[1];
    out[256 + lane] = group_bcast<0>(v); out[320 + lane] = group_bcast<1>(v);
    out[384 + lane] = group_bcast<2>(v); out[448 + lane] = group_bcast<3>(v);
}
int main() {
    int* d; hipMalloc(&d, 512 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"permlane16_swap [0]", "permlane16_swap [1]", "permlane32_swap [0]", "permlane32_swap [1]",
                            "group_bcast<0>", "group_bcast<1>", "group_bcast<2>", "group_bcast<3>"};
    int ok = 1;
    for (int t = 0; t < 8; ++t) {
        printf("%-22s lanes 0,16,32,48 (+3): %6d %6d %6d %6d\n", names[t], h[64 * t + 3], h[64 * t + 19], h[64 * t + 35], h[64 * t + 51]);
        if (t >= 4) for (int l = 0; l < 64; ++l) ok &= (h[64 * t + l] == 1000 * (t - 4) + (l & 15));
    }
    printf("group_bcast correct: %s\n", ok ? "yes" : "NO");
    return ok ? 0 : 1;
}
