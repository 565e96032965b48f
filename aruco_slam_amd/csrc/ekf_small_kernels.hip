// Small per-frame kernels of the EKF update (gfx950):
//   gather : h, dh for every detection + A = H (P + Q)        (k x N, reads 10+3m rows of P)
//   solve  : S = A H^T + R, Cholesky S = L L^T, y = L^-1 (z-h) (one workgroup, LDS resident)
//   panel  : W = L^-1 A, dx = W^T y, state injection           (column chunks)
// plus add_markers / diagonal extraction.
//
// Reference path: extended_kalman_filter.py:95-156 (predict, update),
// :158-237 (parse_poses, landmark_dh), :239-290 (add_marker).
//
// The reference forms K = P H^T S^-1 and P <- (I - K H) P.  With S = L L^T and
// W = L^-1 H P (P symmetric) this is dx = W^T L^-1 (z-h) and P <- P - W^T W.
#include "ekf_kernels.h"
#include "ekf_solve_device.h"

// --------------------------------------------------------------------------
// gather
// --------------------------------------------------------------------------
// NU = detections per wave slot (m <= 4 NU).  All loads of P are issued branch-free
// (clamped indices) before anything consumes them.
//
// One launch, two kinds of workgroups:
//   blockIdx.x <  ncols/64 : A = H (P+Q) for a chunk of 64 columns
//   blockIdx.x >= ncols/64 : one 16x16 block (bi >= bj) of S = Hs (P+Q)[supp,supp] Hs^T + R,
//                            straight from P, written to `sblk` in the column-block layout the
//                            solve kernel keeps in LDS.  S does not wait for A, so the single
//                            workgroup solve kernel starts from a finished S.
// Both first evaluate h / dh for every detection (one thread each) into LDS.
template <typename T, int NU, int MODEL>
__global__ __launch_bounds__(256) void ekf_gather_kernel(EkfFrame fr) {
    constexpr int RD = EkfModel<MODEL>::RD, LMD = EkfModel<MODEL>::LMD, JC = EkfModel<MODEL>::JC;
    constexpr int NSLOT = EKF_CAM + LMD * EkfModel<MODEL>::NDET16;   // P rows one S block can touch
    extern __shared__ __attribute__((aligned(16))) double g_sm[];
    double* hs = g_sm;                                  // [k][JC]
    double* us = g_sm + fr.k * JC;                      // [NSLOT][16]  (S blocks only)
    int* lmc = reinterpret_cast<int*>(us + NSLOT * 16);
    const int tid = threadIdx.x;
    const int m = fr.m, k = fr.k;
    const int nchunk = fr.ncols / 64;
    const bool sblock = (int)blockIdx.x >= nchunk;
    const int chunk0 = blockIdx.x * 64;
    const int cl = tid & 63;
    const int c = sblock ? 0 : chunk0 + cl;        // < ncols <= ld by construction
    const int g = tid >> 6;                        // wave index: wave-uniform landmark
    const T* __restrict__ P = static_cast<const T*>(fr.cov);
    const int64_t ld = fr.ld;
    // camera rows of P for this column and the camera state: independent of idx
    T pcr[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) pcr[a] = P[a * ld + c];
    double cam[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) cam[a] = fr.state[a];
    if (tid < m) lmc[tid] = ekf_lm_column(fr, LMD, tid, blockIdx.x == 0);
    __syncthreads();
    // landmark rows of P for this wave's detections (A chunks)
    T plr[NU][LMD];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int ju = min(g + 4 * u, m - 1);
        const int c0 = lmc[ju];
#pragma unroll
        for (int d = 0; d < LMD; ++d)
            plr[u][d] = P[(int64_t)(c0 + d) * ld + c];
    }
    if (tid < m) {
        const int c0 = lmc[tid];
        double lm[LMD], h[RD], J[RD][JC];
        for (int d = 0; d < LMD; ++d) lm[d] = fr.state[c0 + d];
        ekf_measure_model<MODEL>(cam, lm, h, J);
        for (int d = 0; d < RD; ++d)
            for (int a = 0; a < JC; ++a) hs[(RD * tid + d) * JC + a] = J[d][a];
        if (blockIdx.x == 0) {
            for (int d = 0; d < RD; ++d) {
                for (int a = 0; a < JC; ++a)
                    fr.jac[(size_t)(RD * tid + d) * EKF_JLD + a] = J[d][a];
                fr.resid[RD * tid + d] = fr.z[RD * tid + d] - h[d];   // additive residual, also on q_cl (:140)
            }
            fr.lmcol[tid] = c0;
        }
    }
    __syncthreads();
    if (sblock) {
        // ---- one block of S: rows 16 bi .., columns 16 bj ..
        int sb = blockIdx.x - nchunk, bi = 0;
        while (sb > bi) { sb -= bi + 1; ++bi; }
        const int bj = sb;
        const int j0 = (16 * bi) / RD;                 // first detection that touches the block's rows
        // U[slot][c2] = sum_b Pq[rho(slot)][col(r2, b)] H[r2][b]; rho: slots 0..9 camera rows, slot
        // 10 + LMD (j - j0) + d row d of detection j (<= NDET16 detections touch 16 rows)
        for (int e = tid; e < NSLOT * 16; e += 256) {
            const int slot = e >> 4, c2 = e & 15, r2 = 16 * bj + c2;
            const int j = j0 + (slot - EKF_CAM) / LMD, d = (slot - EKF_CAM) % LMD;
            double acc = 0.0;
            if (r2 < k && (slot < EKF_CAM || j < m)) {
                const int rho = (slot < EKF_CAM) ? slot : lmc[j] + d;
                const T* prw = P + (int64_t)rho * ld;
                const double* h2 = hs + r2 * JC;
                const int c20 = lmc[r2 / RD];
                T pv[JC];
#pragma unroll
                for (int b = 0; b < EKF_CAM; ++b) pv[b] = prw[b];
#pragma unroll
                for (int b = 0; b < LMD; ++b) pv[EKF_CAM + b] = prw[c20 + b];
#pragma unroll
                for (int b = 0; b < JC; ++b) {
                    const int col = (b < EKF_CAM) ? b : c20 + (b - EKF_CAM);
                    const double pq = (double)pv[b] + ((col == rho) ? ekf_qdiag(rho, fr.dims, fr.nz) : 0.0);
                    acc = __builtin_fma(pq, h2[b], acc);
                }
            }
            us[e] = acc;
        }
        __syncthreads();
        double v;
        {
            const int i = tid >> 4, c2 = tid & 15, r1 = 16 * bi + i, r2 = 16 * bj + c2;
            if (r1 >= k || r2 >= k) {
                v = (r1 == r2) ? 1.0 : 0.0;            // identity padding
            } else if (r2 > r1) {
                v = 0.0;                               // strict upper part of a diagonal block: mirrored below
            } else {
                const double* h1 = hs + r1 * JC;
                const int s1 = EKF_CAM + LMD * (r1 / RD - j0);
                double acc = (r1 == r2) ? fr.nz.r_unc : 0.0;
#pragma unroll
                for (int a = 0; a < EKF_CAM; ++a) acc = __builtin_fma(h1[a], us[a * 16 + c2], acc);
#pragma unroll
                for (int d = 0; d < LMD; ++d) acc = __builtin_fma(h1[EKF_CAM + d], us[(s1 + d) * 16 + c2], acc);
                v = acc;
            }
        }
        __syncthreads();                               // `us` is free: it becomes the block's tile
        us[(tid >> 4) * 17 + (tid & 15)] = v;
        __syncthreads();
        if (tid < 64) sv_sblock_emit<false>(fr.sblk + sv_blk_index(bi, bj), us, bi == bj, tid);
        return;
    }
    double pc[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) pc[a] = (double)pcr[a] + ((a == c) ? ekf_qdiag(a, fr.dims, fr.nz) : 0.0);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = g + 4 * u;
        if (j < m) {
            const int c0 = lmc[j];
            double pl[LMD];
#pragma unroll
            for (int d = 0; d < LMD; ++d) pl[d] = (double)plr[u][d] + ((c0 + d == c) ? fr.nz.q_lm : 0.0);
#pragma unroll
            for (int d = 0; d < RD; ++d) {
                const int r = RD * j + d;
                const double* hr = hs + r * JC;
                double acc = 0.0;
#pragma unroll
                for (int a = 0; a < EKF_CAM; ++a) acc = __builtin_fma(hr[a], pc[a], acc);
#pragma unroll
                for (int e = 0; e < LMD; ++e) acc = __builtin_fma(hr[10 + e], pl[e], acc);
                fr.amat[(int64_t)r * fr.lda + c] = acc;
            }
        }
    }
    for (int r = fr.k + g; r < fr.kpad; r += 4) fr.amat[(int64_t)r * fr.lda + c] = 0.0;
}

template <typename T>
void ekf_launch_gather(const EkfFrame& fr, hipStream_t s) {
    const int nb = fr.kpad / EKF_RB;
    const dim3 grid(fr.ncols / 64 + nb * (nb + 1) / 2);
    if (fr.model == 1) {        // EKF_Rotations: 7 rows / detection, m <= 27
        const size_t lds = ((size_t)fr.k * 20 + (EKF_CAM + 10 * 4) * 16) * sizeof(double) + (size_t)fr.m * sizeof(int) + 16;
        if (fr.m <= 32) hipLaunchKernelGGL((ekf_gather_kernel<T, 8, 1>), grid, dim3(256), lds, s, fr);
        else hipLaunchKernelGGL((ekf_gather_kernel<T, 13, 1>), grid, dim3(256), lds, s, fr);      // up to 52 detections
        return;
    }
    const size_t lds = ((size_t)fr.k * EKF_JCOLS + 31 * 16) * sizeof(double) + (size_t)fr.m * sizeof(int) + 16;
    if (fr.m <= 32)
        hipLaunchKernelGGL((ekf_gather_kernel<T, 8, 0>), grid, dim3(256), lds, s, fr);
    else
        hipLaunchKernelGGL((ekf_gather_kernel<T, 16, 0>), grid, dim3(256), lds, s, fr);
}
template void ekf_launch_gather<float>(const EkfFrame&, hipStream_t);
template void ekf_launch_gather<double>(const EkfFrame&, hipStream_t);

// --------------------------------------------------------------------------
// solve
// --------------------------------------------------------------------------
#include "ekf_solve_big.h"

// One workgroup: the shared register-resident factorisation (ekf_solve_device.h: sv_factor) with the
// plain IO policy.  One instantiation per block count NB = kpad / 16.
template <int NB>
__global__ __launch_bounds__(SV_T) void ekf_solve_kernel(EkfFrame fr) {
    extern __shared__ __attribute__((aligned(16))) double v_sm[];
    SvIoPlain io{fr};
    int bad = 0, badcol = 0;
    sv_factor_any<NB>(fr, io, v_sm, bad, badcol);
    if (bad && (threadIdx.x & 63) == 0) ekf_raise(fr, EKF_ST_NOT_SPD);
}

template <int NB>
static void ekf_solve_go(const EkfFrame& fr, hipStream_t s) {
    if (sv_lds_doubles_any(NB) * sizeof(double) > 64 * 1024) {      // > 64 KB of dynamic LDS needs the opt-in
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_solve_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(sv_lds_doubles_any(NB) * sizeof(double)));
            once = true;
        }
    }
    hipLaunchKernelGGL(ekf_solve_kernel<NB>, dim3(1), dim3(SV_T), sv_lds_doubles_any(NB) * sizeof(double), s, fr);
}
void ekf_launch_solve(const EkfFrame& fr, hipStream_t s) {
    switch (fr.kpad / EKF_RB) {
#define SV_CASE(NB) case NB: return ekf_solve_go<NB>(fr, s);
        SV_CASE(1) SV_CASE(2) SV_CASE(3) SV_CASE(4) SV_CASE(5) SV_CASE(6) SV_CASE(7) SV_CASE(8) SV_CASE(9) SV_CASE(10) SV_CASE(11)
        SV_CASE(12) SV_CASE(13) SV_CASE(14) SV_CASE(15) SV_CASE(16) SV_CASE(17) SV_CASE(18) SV_CASE(19) SV_CASE(20) SV_CASE(21)
        SV_CASE(22) SV_CASE(23) SV_CASE(24)
#undef SV_CASE
        default: return;      // (check_config keeps kpad <= 384)
    }
}

// --------------------------------------------------------------------------
// panel
// --------------------------------------------------------------------------
typedef double pf64x4 __attribute__((ext_vector_type(4)));

// MFMA panel kernel: one wave per 16 columns, everything in registers.
// Block row b:  T = A_b - sum_{q<b} L_bq W_q ;  W_b = Dinv_b T.
// W_q sits in the accumulator layout (lane (g,j), reg r <-> row g+4r, col j),
// which IS the B-operand layout of MFMA step r when step r sums over the rows
// {g+4r}; lop/dop hold the A operands in exactly that k order.
template <typename T, int NB>
__global__ __launch_bounds__(64) void ekf_panel_mfma_kernel(EkfFrame fr) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    const int col0 = blockIdx.x * 16;
    const double* __restrict__ lop = fr.lop;
    const double* __restrict__ dop = fr.dop;
    pf64x4 w[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        pf64x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            t[r] = fr.amat[(int64_t)(16 * b + g + 4 * r) * fr.lda + col0 + j];
#pragma unroll
        for (int q = 0; q < b; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double a = lop[(size_t)((b * (b - 1) / 2 + q) * 4 + r) * 64 + lane];
                t = __builtin_amdgcn_mfma_f64_16x16x4f64(a, w[q][r], t, 0, 0, 0);
            }
        }
        pf64x4 wb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double a = dop[(size_t)(b * 4 + r) * 64 + lane];
            wb = __builtin_amdgcn_mfma_f64_16x16x4f64(a, t[r], wb, 0, 0, 0);
        }
        w[b] = wb;
    }
    T* __restrict__ wp = static_cast<T*>(fr.wpanel);
    double part = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * b + g + 4 * r;
            const double v = w[b][r];
            part = __builtin_fma(v, fr.yvec[row], part);
            wp[(int64_t)row * fr.ldw + col0 + j] = (T)v;
            if (fr.wdbg) fr.wdbg[(int64_t)row * fr.ldw + col0 + j] = v;
        }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);                    // dx[col0 + j] in every lane group
    const int col = col0 + j;
    if (fr.model == 1) {            // EKF_Rotations: every landmark has a quaternion -> separate kernel
        if (g == 0 && col < fr.dims) fr.dxvec[col] = part;
        return;
    }
    double nv = 0.0;
    if (g == 0 && (col < 3 || (col >= EKF_CAM && col < fr.dims))) {
        nv = fr.state[col] + part;                   // extended_kalman_filter.py:134-135
        fr.state[col] = nv;
    }
    if (blockIdx.x == 0) {
        const double e0 = __shfl(part, 7), e1 = __shfl(part, 8), e2 = __shfl(part, 9);
        const double x0 = __shfl(nv, 0), x1 = __shfl(nv, 1), x2 = __shfl(nv, 2);
        if (lane == 0) {
            double q[4] = {fr.state[3], fr.state[4], fr.state[5], fr.state[6]};
            const double err[3] = {e0, e1, e2};
            ekf_quat_inject(q, err, fr.quat_mode);
            for (int i = 0; i < 4; ++i) fr.state[3 + i] = q[i];
            for (int i = 0; i < 3; ++i) fr.state[7 + i] = 0.0;   // :152
            if (fr.traj_row) {
                fr.traj_row[0] = x0; fr.traj_row[1] = x1; fr.traj_row[2] = x2;
                for (int i = 0; i < 4; ++i) fr.traj_row[3 + i] = q[i];
            }
        }
    }
}

template <typename T, int NB>
static void ekf_panel_mfma_go(const EkfFrame& fr, hipStream_t s) {
    hipLaunchKernelGGL((ekf_panel_mfma_kernel<T, NB>), dim3(fr.ncols / 16), dim3(64), 0, s, fr);
}

template <typename T>
void ekf_launch_panel(const EkfFrame& fr, hipStream_t s) {
    switch (fr.kpad / EKF_RB) {
#define PN_CASE(NB) case NB: return ekf_panel_mfma_go<T, NB>(fr, s);
        PN_CASE(1) PN_CASE(2) PN_CASE(3) PN_CASE(4) PN_CASE(5) PN_CASE(6) PN_CASE(7) PN_CASE(8) PN_CASE(9) PN_CASE(10) PN_CASE(11)
        PN_CASE(12) PN_CASE(13) PN_CASE(14) PN_CASE(15) PN_CASE(16) PN_CASE(17) PN_CASE(18) PN_CASE(19) PN_CASE(20) PN_CASE(21)
        PN_CASE(22) PN_CASE(23) PN_CASE(24)
#undef PN_CASE
        default: return;
    }
}
template void ekf_launch_panel<float>(const EkfFrame&, hipStream_t);
template void ekf_launch_panel<double>(const EkfFrame&, hipStream_t);

// --------------------------------------------------------------------------
// add_marker (extended_kalman_filter.py:239-290), one thread per new landmark
// --------------------------------------------------------------------------
template <typename T>
__global__ void ekf_add_markers_kernel(T* P, int64_t ld, double* state, int dims,
                                       const double* xyz, const double* unc, double default_unc,
                                       int count) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    double q[4] = {state[3], state[4], state[5], state[6]};
    const double nq = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double a = q[0] * nq, u0 = q[1] * nq, u1 = q[2] * nq, u2 = q[3] * nq;
    const double p[3] = {xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]};
    // rot_cm = R(q)^-1 = R(q)^T  (:264-269);  t_ml = rot_cm p + c  (:272)
    const double d0 = a * a - (u0 * u0 + u1 * u1 + u2 * u2);
    const double up = u0 * p[0] + u1 * p[1] + u2 * p[2];
    const double cx[3] = {u1 * p[2] - u2 * p[1], u2 * p[0] - u0 * p[2], u0 * p[1] - u1 * p[0]};
    const double u[3] = {u0, u1, u2};
    const int c0 = dims + 3 * j;
    for (int d = 0; d < 3; ++d) {
        state[c0 + d] = d0 * p[d] + 2.0 * up * u[d] - 2.0 * a * cx[d] + state[d];
        const double var = unc ? unc[3 * j + d] : default_unc;
        P[(int64_t)(c0 + d) * ld + c0 + d] = (T)var;
    }
}

template <typename T>
void ekf_launch_add_markers(void* cov, int64_t ld, double* state, int32_t dims,
                            const double* xyz_dev, const double* unc_dev, double default_unc,
                            int32_t count, hipStream_t s) {
    hipLaunchKernelGGL(ekf_add_markers_kernel<T>, dim3((count + 63) / 64), dim3(64), 0, s,
                       static_cast<T*>(cov), ld, state, dims, xyz_dev, unc_dev, default_unc, count);
}
template void ekf_launch_add_markers<float>(void*, int64_t, double*, int32_t, const double*,
                                            const double*, double, int32_t, hipStream_t);
template void ekf_launch_add_markers<double>(void*, int64_t, double*, int32_t, const double*,
                                             const double*, double, int32_t, hipStream_t);

// EKF_Rotations state injection (ekf_with_rotations.py:142-177): xyz additive, quaternion
// multiplicative (consistent scalar-first convention) for the camera and for EVERY landmark; the
// landmarks' own error states are never written (they stay 0).  Thread 0 = camera.
__global__ void ekf_inject_rot_kernel(EkfFrame fr, int n_lm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_lm) return;
    const int c0 = (i == 0) ? 0 : EKF_CAM + 10 * (i - 1);
    double* st = fr.state + c0;
    const double* dx = fr.dxvec + c0;
    double q[4] = {st[3], st[4], st[5], st[6]};
    const double err[3] = {dx[7], dx[8], dx[9]};
    ekf_quat_inject(q, err, 1);
    const double x0 = st[0] + dx[0], x1 = st[1] + dx[1], x2 = st[2] + dx[2];
    st[0] = x0; st[1] = x1; st[2] = x2;
    for (int e = 0; e < 4; ++e) st[3 + e] = q[e];
    if (i == 0) {
        for (int e = 0; e < 3; ++e) st[7 + e] = 0.0;         // :157
        if (fr.traj_row) {
            fr.traj_row[0] = x0; fr.traj_row[1] = x1; fr.traj_row[2] = x2;
            for (int e = 0; e < 4; ++e) fr.traj_row[3 + e] = q[e];
        }
    }
}
void ekf_launch_inject_rot(const EkfFrame& fr, int n_lm, hipStream_t s) {
    hipLaunchKernelGGL(ekf_inject_rot_kernel, dim3((n_lm + 1 + 127) / 128), dim3(128), 0, s, fr, n_lm);
}

// EKF_Rotations.add_marker (ekf_with_rotations.py:275-335): pose = [tvec | rvec], rvec read as
// extrinsic xyz Euler angles (:307-310); q_ml = from_matrix(R(q)^-1 R_cl) with SciPy's branch rule.
template <typename T>
__global__ void ekf_add_markers_rot_kernel(T* P, int64_t ld, double* state, int dims,
                                           const double* pose6, const double* unc, double default_unc,
                                           int count) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    double q[4] = {state[3], state[4], state[5], state[6]};
    const double nq = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double a = q[0] * nq, u0 = q[1] * nq, u1 = q[2] * nq, u2 = q[3] * nq;
    // rot_cm = R(q)^T
    const double rcm[3][3] = {
        {a * a + u0 * u0 - u1 * u1 - u2 * u2, 2 * (u0 * u1 + a * u2), 2 * (u0 * u2 - a * u1)},
        {2 * (u0 * u1 - a * u2), a * a - u0 * u0 + u1 * u1 - u2 * u2, 2 * (u1 * u2 + a * u0)},
        {2 * (u0 * u2 + a * u1), 2 * (u1 * u2 - a * u0), a * a - u0 * u0 - u1 * u1 + u2 * u2}};
    const double* ps = pose6 + 6 * j;
    const double ca = cos(ps[3]), sa = sin(ps[3]), cb = cos(ps[4]), sb = sin(ps[4]), cc = cos(ps[5]), sc = sin(ps[5]);
    // R_cl = Rz(c) Ry(b) Rx(a)
    const double rcl[3][3] = {{cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa},
                              {sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa},
                              {-sb, cb * sa, cb * ca}};
    double mm[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c2 = 0; c2 < 3; ++c2)
            mm[r][c2] = rcm[r][0] * rcl[0][c2] + rcm[r][1] * rcl[1][c2] + rcm[r][2] * rcl[2][c2];
    // matrix -> quaternion (x y z w), branch on the largest of (M00, M11, M22, trace)
    const double tr = mm[0][0] + mm[1][1] + mm[2][2];
    double dec[4] = {mm[0][0], mm[1][1], mm[2][2], tr};
    int ch = 0;
    for (int e = 1; e < 4; ++e)
        if (dec[e] > dec[ch]) ch = e;
    double qx[4];
    if (ch != 3) {
        const int i2 = ch, j2 = (ch + 1) % 3, k2 = (ch + 2) % 3;
        qx[i2] = 1.0 - tr + 2.0 * mm[i2][i2];
        qx[j2] = mm[j2][i2] + mm[i2][j2];
        qx[k2] = mm[k2][i2] + mm[i2][k2];
        qx[3] = mm[k2][j2] - mm[j2][k2];
    } else {
        qx[0] = mm[2][1] - mm[1][2];
        qx[1] = mm[0][2] - mm[2][0];
        qx[2] = mm[1][0] - mm[0][1];
        qx[3] = 1.0 + tr;
    }
    const double nn = 1.0 / sqrt(qx[0] * qx[0] + qx[1] * qx[1] + qx[2] * qx[2] + qx[3] * qx[3]);
    const int c0 = dims + 10 * j;
    for (int d = 0; d < 3; ++d)
        state[c0 + d] = rcm[d][0] * ps[0] + rcm[d][1] * ps[1] + rcm[d][2] * ps[2] + state[d];
    state[c0 + 3] = qx[3] * nn;
    state[c0 + 4] = qx[0] * nn;
    state[c0 + 5] = qx[1] * nn;
    state[c0 + 6] = qx[2] * nn;
    for (int d = 7; d < 10; ++d) state[c0 + d] = 0.0;
    for (int d = 0; d < 10; ++d) {
        const double var = unc ? unc[10 * j + d] : default_unc;
        P[(int64_t)(c0 + d) * ld + c0 + d] = (T)var;
    }
}
template <typename T>
void ekf_launch_add_markers_rot(void* cov, int64_t ld, double* state, int32_t dims, const double* pose6_dev,
                                const double* unc_dev, double default_unc, int32_t count, hipStream_t s) {
    hipLaunchKernelGGL(ekf_add_markers_rot_kernel<T>, dim3((count + 63) / 64), dim3(64), 0, s,
                       static_cast<T*>(cov), ld, state, dims, pose6_dev, unc_dev, default_unc, count);
}
template void ekf_launch_add_markers_rot<float>(void*, int64_t, double*, int32_t, const double*,
                                                const double*, double, int32_t, hipStream_t);
template void ekf_launch_add_markers_rot<double>(void*, int64_t, double*, int32_t, const double*,
                                                 const double*, double, int32_t, hipStream_t);

template <typename T>
__global__ void ekf_cov_diag_kernel(const T* P, int64_t ld, double* out, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (double)P[(int64_t)i * ld + i];
}
template <typename T>
void ekf_launch_cov_diag(const void* cov, int64_t ld, double* out_dev, int32_t count, hipStream_t s) {
    hipLaunchKernelGGL(ekf_cov_diag_kernel<T>, dim3((count + 255) / 256), dim3(256), 0, s,
                       static_cast<const T*>(cov), ld, out_dev, count);
}
template void ekf_launch_cov_diag<float>(const void*, int64_t, double*, int32_t, hipStream_t);
template void ekf_launch_cov_diag<double>(const void*, int64_t, double*, int32_t, hipStream_t);
