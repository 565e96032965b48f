// Probe (MI355X): cost of a cross-stream dependency per hop, three mechanisms.  Chain: kernel on stream A ->
// kernel on stream B -> kernel on stream A -> ... (each kernel: one workgroup, ~2 us of spinning), 200 hops.
//   0 one stream (reference: no cross-stream edge)
//   1 hipEventRecord + hipStreamWaitEvent
//   2 hipStreamWriteValue64 + hipStreamWaitValue64 (>=)
//   3 device-side: the consumer kernel itself spins on a counter that the producer kernel bumps when it is done
//     (both kernels are enqueued ahead of time; the consumer occupies one wave while it waits)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/xstream_probe.hip -o aruco_slam_amd/build/xstream_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void work(unsigned long long* ctr, unsigned long long wait_for, long long ticks) {
    if (wait_for) {     // mechanism 3: wait until the previous hop has finished
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < wait_for) __builtin_amdgcn_s_sleep(2);
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
int main() {
    unsigned long long* ctr; hipMalloc(&ctr, 64);
    uint64_t* sig; hipMalloc(&sig, 64);
    hipStream_t s[2]; hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
    const int hops = 200;
    std::vector<hipEvent_t> ev(hops);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const char* names[4] = {"one stream", "event record + stream wait event", "stream write value + wait value", "consumer kernel spins on a device counter"};
    for (int mech = 0; mech < 4; ++mech)
        for (int trial = 0; trial < 2; ++trial) {
            hipMemset(ctr, 0, 64); hipMemset(sig, 0, 64);
            hipDeviceSynchronize();
            auto w0 = std::chrono::steady_clock::now();
            for (int h = 0; h < hops; ++h) {
                hipStream_t cur = mech == 0 ? s[0] : s[h & 1], nxt = mech == 0 ? s[0] : s[(h + 1) & 1];
                hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, cur, ctr, (mech == 3 && h > 0) ? (unsigned long long)h : 0ull, 200LL);   // 2 us
                if (mech == 1) { hipEventRecord(ev[h], cur); hipStreamWaitEvent(nxt, ev[h], 0); }
                if (mech == 2) { hipStreamWriteValue64(cur, sig, (uint64_t)(h + 1), 0); hipStreamWaitValue64(nxt, sig, (uint64_t)(h + 1), hipStreamWaitValueGte, ~0ull); }
            }
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
            if (trial) printf("%-46s: %7.2f us per hop (2 us of it is the kernel's own spin)\n", names[mech], us / hops);
        }
    return 0;
}
