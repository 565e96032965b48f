// Covariance update  P <- P + Q - W^T W   (the dominant kernel of the path).
//
// Reference: P <- (I - K H) P after P += Q (extended_kalman_filter.py:95-105,
// 155-156), a dense N^3 GEMM there.  With S = L L^T and W = L^-1 H (P+Q)
// (k x N, k = 3 m) the same update is the symmetric rank-k downdate below:
// 2 N^2 k flops, P read once and written once (2 N^2 sizeof(T) bytes).
//
// W is a k-major panel [kpad][ldw]; rows >= k and columns >= N are zero, so
// the padded part of P stays exactly zero without any masking.
//
// Bitwise symmetry: element (i,j) and (j,i) run the same k-ordered fma chain
// on commuting products, so P stays exactly symmetric.
//
//  * VALU kernel : plain LDS-tiled reference implementation (any T), computes
//                  every tile of the full matrix.
//  * MFMA kernels: one wavefront per 32x32 tile of the LOWER triangle (I >= J).
//                  No LDS staging and no barriers: the k-major W panel is tiny
//                  (k x N) and L2 resident, so each lane pulls its two operands
//                  per MFMA straight from it.  f32: the operands go through a ring
//                  of 3 register slots of 8 k-pairs (inline-asm loads, hand-counted
//                  waits), < 96 registers per wave, so 5 waves per SIMD (4 above
//                  k = 128) are resident and every tile of the n=1024 problem is in
//                  flight from the start; the hardware overlaps one wave's P traffic
//                  with other waves' matrix work.  An off-diagonal tile is computed
//                  once and written twice: D to (I,J) and, transposed through a
//                  wave-private LDS tile, D^T to (J,I).  Flops are halved
//                  (N^2 k), P is read once per lower tile and written once.
//                  f32: v_mfma_f32_32x32x2_f32, one accumulator.
//                  f64: v_mfma_f64_16x16x4_f64, 2x2 accumulators (one workgroup per
//                  tile / one sub-tile per wave for small problems).
#include <hip/hip_ext.h>
#include "ekf_kernels.h"

// Stores of the updated covariance.  P is not read again inside the launch and every XCD's L2 is invalidated before the
// next launch reads it, so keeping the 38.5 MB (n=1024) of written lines dirty in the L2s only postpones their way to
// memory to the end of the launch, where the next kernel waits for it.  Write-through (sc1) stores stream them out while
// the kernel still computes (tools/store_mode_probe.sh, n=1024, m=32, f32; kernel / serial frame / pipelined frame in us:
// plain 16.7 / 40.0 / 36.1 - nontemporal 15.9 / 39.2 / 35.2 - write-through 15.2 / 38.1 / 34.6).
// EKF_COV_STORE_MODE (experiments): 0 plain, 1 nontemporal, 2 write-through.
#ifndef EKF_COV_STORE_MODE
#define EKF_COV_STORE_MODE 2
#endif
template <typename T>
__device__ __forceinline__ void ekf_cov_store(T* p, T v) {
    if (EKF_COV_STORE_MODE == 1) __builtin_nontemporal_store(v, p);
    else if (EKF_COV_STORE_MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// --------------------------------------------------------------------------
// VALU reference kernel: 64x64 tile, 256 threads, 4x4 per thread
// --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ekf_cov_update_valu(EkfFrame fr) {
    __shared__ T sa[16][64];
    __shared__ T sb[16][64];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const T* __restrict__ wp = static_cast<const T*>(fr.wpanel);
    T acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = (T)0;
    for (int kc = 0; kc < fr.kpad; kc += 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + 256 * t, r = e >> 6, cc = e & 63;
            sa[r][cc] = wp[(int64_t)(kc + r) * fr.ldw + i0 + cc];
            sb[r][cc] = wp[(int64_t)(kc + r) * fr.ldw + j0 + cc];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            T a[4], b[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) a[x] = -sa[r][ty * 4 + x];
#pragma unroll
            for (int y = 0; y < 4; ++y) b[y] = sb[r][tx * 4 + y];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_fma(a[x], b[y], acc[x][y]);
        }
        __syncthreads();
    }
    T* __restrict__ P = static_cast<T*>(fr.cov);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int row = i0 + ty * 4 + x;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int col = j0 + tx * 4 + y;
            T v = P[(int64_t)row * fr.ld + col];
            if (row == col) v += (T)ekf_qdiag(row, fr.dims, fr.nz);
            P[(int64_t)row * fr.ld + col] = v + acc[x][y];
        }
    }
}

// --------------------------------------------------------------------------
// tile index -> (I, J), I >= J, row-major over the lower triangle
// --------------------------------------------------------------------------
__device__ __forceinline__ void ekf_tri_decode(int item, int& I, int& J) {
    int i = (int)((sqrtf(8.0f * (float)item + 1.0f) - 1.0f) * 0.5f);
    while (i * (i + 1) / 2 > item) --i;
    while ((i + 1) * (i + 2) / 2 <= item) ++i;
    I = i;
    J = item - i * (i + 1) / 2;
}

// --------------------------------------------------------------------------
// MFMA f32: one wave per 32x32 lower tile, v_mfma_f32_32x32x2_f32 over k in ascending order.
// Both operands come straight from the L2-resident k-major W panel (no LDS staging, no barriers)
// through a ring of D = 3 register slots of 8 k-pairs, so a wave needs < 96 registers and FIVE
// waves fit a SIMD: every tile of the n=1024 problem (18.6 per CU) is resident from the start and
// the launch runs as one round (the earlier version held all 96 operands in registers: 132
// registers, 3 waves per SIMD, ~1.3 rounds, 19.3 us instead of 15.9).
// --------------------------------------------------------------------------
template <int KB, int DMAX = 3, int WPE = 5>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void ekf_cov_update_mfma_f32(EkfFrame fr, int nitems) {
    __shared__ float tr[4][32][33];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (item >= nitems) return;
    int I, J;
    ekf_tri_decode(item, I, J);
    const int i0 = 32 * I, j0 = 32 * J;
    const float* __restrict__ wp = static_cast<const float*>(fr.wpanel);
    // pipelined sequence mode: P_t is read from `cov` and P_{t+1} is written to `cov_out` (the next frame's front
    // kernel reads P_t beside this launch); otherwise in place
    const float* __restrict__ P = static_cast<const float*>(fr.cov);
    float* Pout = fr.cov_out ? static_cast<float*>(fr.cov_out) : static_cast<float*>(fr.cov);
    const int ld = (int)fr.ld, ldw = (int)fr.ldw;
    constexpr int NCH = KB, D = (NCH < DMAX) ? NCH : DMAX;
    float ra[D][8], rb[D][8], pt[16];
    // Loads are written as inline asm (scalar base + one 32-bit lane offset, no 64-bit address
    // registers) with hand-counted s_waitcnt: hipcc's own allocation at this register budget spills
    // freshly loaded operands, which serialises the ring.
    // Issue order: W chunks 0 .. D-1, the P tile (the HBM stream starts at once, but the first MFMAs
    // do not wait for it), then chunk c + D after the MFMAs of chunk c.
    const unsigned wlane = 4u * (unsigned)(lhi * ldw + l31);        // A[i = l31][k = lhi], B[k = lhi][j = l31]
    const unsigned plane = 4u * (unsigned)(4 * lhi * ld + l31);     // C/D layout: row (reg&3) + 8 (reg>>2) + 4 lhi
    const uint64_t kstep = 8ull * (uint64_t)ldw;                    // two rows of W in bytes
    uint64_t ua = reinterpret_cast<uint64_t>(wp + i0), ub = reinterpret_cast<uint64_t>(wp + j0);
    const float* __restrict__ ptile = P + (int64_t)i0 * ld + j0;
    float* otile = Pout + (int64_t)i0 * ld + j0;
#define EKF_GLD(dst, voff, sbase) asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")
#define EKF_WAIT16(n, A, B)                                                                              \
    asm volatile("s_waitcnt vmcnt(%16)"                                                                   \
                 : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]), \
                   "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), "+v"(B[4]), "+v"(B[5]), "+v"(B[6]), "+v"(B[7])  \
                 : "n"(n))
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            EKF_GLD(ra[c][u], wlane, ua);
            EKF_GLD(rb[c][u], wlane, ub);
            ua += kstep;
            ub += kstep;
        }
    }
    {
        uint64_t pb = reinterpret_cast<uint64_t>(ptile);
        const uint64_t r1 = 4ull * (uint64_t)ld, r5 = 20ull * (uint64_t)ld;   // bytes: +1 row, +5 rows
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {         // rows (reg&3) + 8 (reg>>2)
            EKF_GLD(pt[reg], plane, pb);
            pb += ((reg & 3) == 3) ? r5 : r1;
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        // loads that may still be in flight when chunk c is needed (issued after it)
        const int later = (c < NCH - D) ? c : NCH - D;                           // chunks D .. issued so far
        const int allow = (c < D) ? 16 * (D - c - 1) + 16 + 16 * later : 16 * (D + later - c - 1);
        switch (allow) {      // (immediate operand; the counter saturates at 63)
            case 0: EKF_WAIT16(0, ra[c % D], rb[c % D]); break;
            case 16: EKF_WAIT16(16, ra[c % D], rb[c % D]); break;
            case 32: EKF_WAIT16(32, ra[c % D], rb[c % D]); break;
            case 48: EKF_WAIT16(48, ra[c % D], rb[c % D]); break;
            default: EKF_WAIT16(63, ra[c % D], rb[c % D]); break;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-ra[c % D][u], rb[c % D][u], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + D < NCH) {                       // next chunk into the slot just freed
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                EKF_GLD(ra[c % D][u], wlane, ua);
                EKF_GLD(rb[c % D][u], wlane, ub);
                ua += kstep;
                ub += kstep;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(pt[0]), "+v"(pt[1]), "+v"(pt[2]), "+v"(pt[3]), "+v"(pt[4]), "+v"(pt[5]), "+v"(pt[6]), "+v"(pt[7]),
                   "+v"(pt[8]), "+v"(pt[9]), "+v"(pt[10]), "+v"(pt[11]), "+v"(pt[12]), "+v"(pt[13]), "+v"(pt[14]), "+v"(pt[15]));
#undef EKF_GLD
#undef EKF_WAIT16
    const float qlane = (I == J) ? (float)ekf_qdiag(i0 + l31, fr.dims, fr.nz) : 0.0f;
    float out[16];
    auto stf = [](float* base, unsigned off, float v) { ekf_cov_store(reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off), v); };
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rl = (reg & 3) + 8 * (reg >> 2) + 4 * lhi;
        const float v = (pt[reg] + ((rl == l31) ? qlane : 0.0f)) + acc[reg];
        out[reg] = v;
        stf(otile + (int64_t)((reg & 3) + 8 * (reg >> 2)) * ld, plane, v);
    }
    if (I != J) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            tr[wave][(reg & 3) + 8 * (reg >> 2) + 4 * lhi][l31] = out[reg];
        __builtin_amdgcn_wave_barrier();
        float* pm = Pout + (int64_t)j0 * ld + i0;
        const unsigned mlane = 4u * (unsigned)(lhi * ld + l31);
#pragma unroll
        for (int it = 0; it < 16; ++it)          // column 2 it + lhi of D = row of D^T
            stf(pm + (int64_t)(2 * it) * ld, mlane, tr[wave][l31][2 * it + lhi]);
    }
}

// --------------------------------------------------------------------------
// MFMA f64: one wave per 32x32 lower tile = 2x2 tiles of 16x16x4
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ekf_cov_update_mfma_f64(EkfFrame fr, int nitems) {
    __shared__ double tr[4][32][33];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int item = blockIdx.x * 4 + wave;
    if (item >= nitems) return;
    int I, J;
    ekf_tri_decode(item, I, J);
    const int i0 = 32 * I, j0 = 32 * J;
    const double* __restrict__ wp = static_cast<const double*>(fr.wpanel);
    const double* P = static_cast<const double*>(fr.cov);      // (pipelined sequence mode: P_t in, P_{t+1} out elsewhere)
    double* Pout = fr.cov_out ? static_cast<double*>(fr.cov_out) : static_cast<double*>(fr.cov);
    const int64_t ld = fr.ld, ldw = fr.ldw;

    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 reg
    double pt[2][2][4];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pt[ri][ci][r] = P[(int64_t)(i0 + 16 * ri + g + 4 * r) * ld + j0 + 16 * ci + c];
    f64x4 acc[2][2];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ri][ci][r] = 0.0;
    // A[i = c][k = g] = -W[k][i0 + 16 ri + c],  B[k = g][j = c] = W[k][j0 + 16 ci + c]
    const double* wa = wp + (int64_t)g * ldw + i0 + c;
    const double* wb = wp + (int64_t)g * ldw + j0 + c;
    const int steps = fr.kpad >> 2;                   // multiple of 4
    for (int ks = 0; ks < steps; ks += 4) {
        double a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t off = (int64_t)(4 * (ks + u)) * ldw;
            a0[u] = wa[off];
            a1[u] = wa[off + 16];
            b0[u] = wb[off];
            b1[u] = wb[off + 16];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[u], b0[u], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[u], b1[u], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[u], b0[u], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[u], b1[u], acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 16 * ri + g + 4 * r, cl = 16 * ci + c;
                double v = pt[ri][ci][r];
                if (I == J && rl == cl) v += ekf_qdiag(i0 + rl, fr.dims, fr.nz);
                v += acc[ri][ci][r];
                ekf_cov_store(Pout + (int64_t)(i0 + rl) * ld + j0 + cl, v);
                if (I != J) tr[wave][rl][cl] = v;
            }
    if (I != J) {
        __builtin_amdgcn_wave_barrier();
        const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int cc = 2 * it + lhi;
            ekf_cov_store(Pout + (int64_t)(j0 + cc) * ld + i0 + l31, tr[wave][l31][cc]);
        }
    }
}

// --------------------------------------------------------------------------
// MFMA f64, small problems: the same 32x32 tiles, but ONE WORKGROUP per tile and one 16x16 sub-tile
// per wave (same MFMA sequence per sub-tile: bitwise the results of the kernel above).  With a few
// hundred tiles the one-wave-per-tile kernel leaves most of the 1024 SIMDs idle and every wave runs
// 4x the serial work; here n=256 (325 tiles) becomes 1300 waves.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ekf_cov_update_mfma_f64_split(EkfFrame fr, int nitems) {
    __shared__ double tr[32][33];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int ri = wave >> 1, ci = wave & 1;
    int I, J;
    ekf_tri_decode(blockIdx.x, I, J);
    const int i0 = 32 * I, j0 = 32 * J;
    const double* __restrict__ wp = static_cast<const double*>(fr.wpanel);
    const double* P = static_cast<const double*>(fr.cov);
    double* Pout = fr.cov_out ? static_cast<double*>(fr.cov_out) : static_cast<double*>(fr.cov);
    const int64_t ld = fr.ld, ldw = fr.ldw;
    double pt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pt[r] = P[(int64_t)(i0 + 16 * ri + g + 4 * r) * ld + j0 + 16 * ci + c];
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const double* wa = wp + (int64_t)g * ldw + i0 + 16 * ri + c;
    const double* wb = wp + (int64_t)g * ldw + j0 + 16 * ci + c;
    const int steps = fr.kpad >> 2;                   // multiple of 4
    for (int ks = 0; ks < steps; ks += 4) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t off = (int64_t)(4 * (ks + u)) * ldw;
            a[u] = wa[off];
            b[u] = wb[off];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rl = 16 * ri + g + 4 * r, cl = 16 * ci + c;
        double v = pt[r];
        if (I == J && rl == cl) v += ekf_qdiag(i0 + rl, fr.dims, fr.nz);
        v += acc[r];
        ekf_cov_store(Pout + (int64_t)(i0 + rl) * ld + j0 + cl, v);
        if (I != J) tr[rl][cl] = v;
    }
    if (I != J) {
        __syncthreads();
        const int l31 = threadIdx.x & 31;
        for (int cc = threadIdx.x >> 5; cc < 32; cc += 8)          // column cc of D = row of D^T
            ekf_cov_store(Pout + (int64_t)(j0 + cc) * ld + i0 + l31, tr[l31][cc]);
    }
}

// Pipelined sequence mode: ordering between the two streams on the device (an event pair costs ~13 us per edge,
// stream write / wait values ~7 us, a resident one-wave kernel that polls a counter ~1.2 us:
// tools/probes/xstream_probe.hip).  The gate keeps whatever follows it on its stream from starting before
// `counter >= target`; the wait is bounded (status bit instead of a hang).
__global__ void ekf_gate_kernel(unsigned long long* counter, unsigned long long target, int32_t* status, int max_polls) {
    if (threadIdx.x != 0) return;
    int it = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
        if (++it > max_polls) { atomicOr(status, EKF_ST_GATE_TIMEOUT); break; }
        __builtin_amdgcn_s_sleep(8);
    }
}
__global__ void ekf_signal_kernel(unsigned long long* counter, unsigned long long value) {
    if (threadIdx.x == 0) __hip_atomic_store(counter, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void ekf_launch_gate(unsigned long long* counter, unsigned long long target, int32_t* status, hipStream_t s, int max_polls) {
    hipLaunchKernelGGL(ekf_gate_kernel, dim3(1), dim3(64), 0, s, counter, target, status, max_polls);
}
void ekf_launch_signal(unsigned long long* counter, unsigned long long value, hipStream_t s) {
    hipLaunchKernelGGL(ekf_signal_kernel, dim3(1), dim3(64), 0, s, counter, value);
}

// plain launch; with events attached, the dispatch's own start / stop time stamps land in e0 / e1
#define EKF_COV_LAUNCH(kernel, grid, block, s, e0, e1, ...)                                     \
    do {                                                                                         \
        if (e0) hipExtLaunchKernelGGL(kernel, grid, block, 0, s, e0, e1, 0, __VA_ARGS__);        \
        else hipLaunchKernelGGL(kernel, grid, block, 0, s, __VA_ARGS__);                         \
    } while (0)

static inline int ekf_tri_items(const EkfFrame& fr) {
    const int t = (fr.dims + 31) / 32;                // 32 t <= ncols <= ld
    return t * (t + 1) / 2;
}

template <>
void ekf_launch_cov_update<float>(const EkfFrame& fr, int variant, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (variant == 1) {
        EKF_COV_LAUNCH(ekf_cov_update_valu<float>, dim3(fr.ncols / 64, fr.ncols / 64), dim3(256), s, e0, e1, fr);
    } else if (fr.cov_tiles) {
        ekf_launch_cov_update_macro(fr, s, e0, e1);      // large problems: one workgroup per 128 x 128 macro tile
    } else {
        const int items = ekf_tri_items(fr);
        const dim3 grid((items + 3) / 4), block(256);
        switch (fr.kpad / 16) {
            // The kernel counts its own loads (s_waitcnt by hand), so it must not spill: 5 waves per SIMD
            // (96 registers) up to k = 128, 4 waves (128 registers) above, where hipcc needs a few more
            // (tests/test_host_cpu.py checks the spill count of every instantiation).
#define EKF_COV_CASE(KB) case KB: EKF_COV_LAUNCH((ekf_cov_update_mfma_f32<KB, 3, (KB <= 8) ? 5 : 4>), grid, block, s, e0, e1, fr, items); break;
            EKF_COV_CASE(1) EKF_COV_CASE(2) EKF_COV_CASE(3) EKF_COV_CASE(4) EKF_COV_CASE(5) EKF_COV_CASE(6)
            EKF_COV_CASE(7) EKF_COV_CASE(8) EKF_COV_CASE(9) EKF_COV_CASE(10) EKF_COV_CASE(11) EKF_COV_CASE(12)
            EKF_COV_CASE(13) EKF_COV_CASE(14) EKF_COV_CASE(15) EKF_COV_CASE(16) EKF_COV_CASE(17) EKF_COV_CASE(18)
            EKF_COV_CASE(19) EKF_COV_CASE(20) EKF_COV_CASE(21) EKF_COV_CASE(22) EKF_COV_CASE(23) EKF_COV_CASE(24)
            default: break;      // (check_config keeps kpad <= 384)
#undef EKF_COV_CASE
        }
    }
}
template <>
void ekf_launch_cov_update<double>(const EkfFrame& fr, int variant, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (variant == 1) {
        EKF_COV_LAUNCH(ekf_cov_update_valu<double>, dim3(fr.ncols / 64, fr.ncols / 64), dim3(256), s, e0, e1, fr);
    } else {
        const int items = ekf_tri_items(fr);
        if (items <= 2048)      // small problem: one workgroup per tile, one 16x16 sub-tile per wave
            EKF_COV_LAUNCH(ekf_cov_update_mfma_f64_split, dim3(items), dim3(256), s, e0, e1, fr, items);
        else
            EKF_COV_LAUNCH(ekf_cov_update_mfma_f64, dim3((items + 3) / 4), dim3(256), s, e0, e1, fr, items);
    }
}
