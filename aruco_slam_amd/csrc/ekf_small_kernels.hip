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
#include <cstdlib>

#include "ekf_kernels.h"

// --------------------------------------------------------------------------
// gather
// --------------------------------------------------------------------------
// NU = detections per wave slot (m <= 4 NU).  All loads of P are issued branch-free
// (clamped indices) before anything consumes them.
template <typename T, int NU>
__global__ __launch_bounds__(256) void ekf_gather_kernel(EkfFrame fr) {
    extern __shared__ __attribute__((aligned(16))) double g_sm[];
    __shared__ int scount[64];           // support slots that live in this column chunk
    __shared__ int sslot[64][4];
    double* hs = g_sm;                                  // [k][13]
    int* lmc = reinterpret_cast<int*>(g_sm + fr.k * EKF_JCOLS);
    const int tid = threadIdx.x;
    const int m = fr.m;
    const int chunk0 = blockIdx.x * 64;
    const int cl = tid & 63;
    const int c = chunk0 + cl;                     // < ncols <= ld by construction
    const int g = tid >> 6;                        // wave index: wave-uniform landmark
    const T* __restrict__ P = static_cast<const T*>(fr.cov);
    const int64_t ld = fr.ld;
    // camera rows of P for this column and the camera state: independent of idx
    // support rows come from P, or -- cross-frame lookahead -- from the priority-row buffer that
    // ekf_cov_rows_kernel filled with exactly the values the covariance update is writing into P
    const T* __restrict__ prow = static_cast<const T*>(fr.prow);
    T pcr[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) pcr[a] = prow ? prow[(int64_t)a * fr.ldw + c] : P[a * ld + c];
    double cam[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) cam[a] = fr.state[a];
    if (tid < 64) scount[tid] = 0;
    if (tid < m) lmc[tid] = EKF_CAM + EKF_LM * fr.idx[tid];
    __syncthreads();
    // landmark rows of P for this wave's detections
    T plr[NU][3];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int ju = min(g + 4 * u, m - 1);
        const int c0 = lmc[ju];
#pragma unroll
        for (int d = 0; d < 3; ++d)
            plr[u][d] = prow ? prow[(int64_t)(EKF_CAM + 3 * ju + d) * fr.ldw + c] : P[(int64_t)(c0 + d) * ld + c];
    }
    if (tid < m) {
        const int c0 = lmc[tid];
        double lm[3], h[3], J[3][EKF_JCOLS];
        for (int d = 0; d < 3; ++d) lm[d] = fr.state[c0 + d];
        ekf_measure(cam, lm, h, J);
        for (int d = 0; d < 3; ++d)
            for (int a = 0; a < EKF_JCOLS; ++a) hs[(3 * tid + d) * EKF_JCOLS + a] = J[d][a];
        for (int e = 0; e < 3; ++e) {
            const int sl = c0 + e - chunk0;
            if (sl >= 0 && sl < 64) {
                const int p = atomicAdd(&scount[sl], 1);
                if (p < 4) sslot[sl][p] = 3 * tid + e;
            }
        }
        if (blockIdx.x == 0) {
            for (int d = 0; d < 3; ++d) {
                for (int a = 0; a < EKF_JCOLS; ++a)
                    fr.jac[(size_t)(3 * tid + d) * EKF_JLD + a] = J[d][a];
                fr.resid[3 * tid + d] = fr.z[3 * tid + d] - h[d];
            }
            fr.lmcol[tid] = c0;
        }
    }
    __syncthreads();
    double pc[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) pc[a] = (double)pcr[a] + ((a == c) ? ekf_qdiag(a, fr.dims, fr.nz) : 0.0);
    const int nslot = scount[cl];
    int slots[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) slots[q] = sslot[cl][q];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = g + 4 * u;
        if (j < m) {
            const int c0 = lmc[j];
            double pl[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) pl[d] = (double)plr[u][d] + ((c0 + d == c) ? fr.nz.q_lm : 0.0);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int r = 3 * j + d;
                const double* hr = hs + r * EKF_JCOLS;
                double acc = 0.0;
#pragma unroll
                for (int a = 0; a < EKF_CAM; ++a) acc += hr[a] * pc[a];
#pragma unroll
                for (int e = 0; e < 3; ++e) acc += hr[10 + e] * pl[e];
                fr.amat[(int64_t)r * fr.lda + c] = acc;
                if (nslot > 0) {
                    if (nslot <= 4) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (q < nslot) {
                                fr.asup[(size_t)r * fr.ldas + slots[q]] = acc;
                                fr.asupt[(size_t)slots[q] * fr.ldast + r] = acc;
                            }
                    } else {                       // > 4 detections of one landmark in a frame
                        for (int jp = 0; jp < m; ++jp)
                            for (int e = 0; e < 3; ++e)
                                if (lmc[jp] + e == c) {
                                    fr.asup[(size_t)r * fr.ldas + 3 * jp + e] = acc;
                                    fr.asupt[(size_t)(3 * jp + e) * fr.ldast + r] = acc;
                                }
                    }
                }
            }
        }
    }
    for (int r = fr.k + g; r < fr.kpad; r += 4) fr.amat[(int64_t)r * fr.lda + c] = 0.0;
}

template <typename T>
void ekf_launch_gather(const EkfFrame& fr, hipStream_t s) {
    const size_t lds = (size_t)fr.k * EKF_JCOLS * sizeof(double) + (size_t)fr.m * sizeof(int) + 16;
    if (fr.m <= 32)
        hipLaunchKernelGGL((ekf_gather_kernel<T, 8>), dim3(fr.ncols / 64), dim3(256), lds, s, fr);
    else
        hipLaunchKernelGGL((ekf_gather_kernel<T, 16>), dim3(fr.ncols / 64), dim3(256), lds, s, fr);
}
template void ekf_launch_gather<float>(const EkfFrame&, hipStream_t);
template void ekf_launch_gather<double>(const EkfFrame&, hipStream_t);

// --------------------------------------------------------------------------
// solve: one workgroup.  S (lower, packed) and the augmented residual row live
// in LDS: (kpad+1)(kpad+2)/2 + kpad doubles  (kpad = 192 -> 151 KB of 160 KB).
// --------------------------------------------------------------------------
int ekf_solve_lds_bytes(int kpad) {
    return (int)(((size_t)(kpad + 1) * (kpad + 2) / 2 + kpad) * sizeof(double));
}

__global__ __launch_bounds__(256) void ekf_solve_kernel(EkfFrame fr) {
    extern __shared__ __attribute__((aligned(16))) double s_sm[];
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int k = fr.k, kp = fr.kpad;
    double* S = s_sm;
    double* diag = s_sm + (size_t)(kp + 1) * (kp + 2) / 2;
#define SP(i, j) S[((i) * ((i) + 1)) / 2 + (j)]
    // S = A[:, support] H^T + R  (lower triangle); row kp = residual (augmented:
    // its factor row is y = L^-1 (z - h)); rows k..kp-1 = identity padding.
    for (int i = ti; i <= kp; i += 16) {
        for (int j = tj; j <= i && j < kp; j += 16) {
            double v;
            if (i == kp) {
                v = (j < k) ? fr.resid[j] : 0.0;
            } else if (i >= k) {
                v = (i == j) ? 1.0 : 0.0;
            } else {
                const double* hj = fr.jac + (size_t)j * EKF_JLD;
                const double* ai = fr.amat + (int64_t)i * fr.lda;
                const int c0 = fr.lmcol[j / 3];
                double acc = (i == j) ? fr.nz.r_unc : 0.0;
#pragma unroll
                for (int b = 0; b < EKF_CAM; ++b) acc += ai[b] * hj[b];
#pragma unroll
                for (int e = 0; e < 3; ++e) acc += ai[c0 + e] * hj[10 + e];
                v = acc;
            }
            SP(i, j) = v;
        }
    }
    __syncthreads();
    // right-looking Cholesky, column by column
    for (int j = 0; j < kp; ++j) {
        const double d = SP(j, j);
        if (tid == 0) {
            if (!(d > 0.0)) atomicOr(fr.status, 1);
            diag[j] = sqrt(d);
        }
        const double rs = 1.0 / sqrt(d);
        for (int i = j + 1 + tid; i <= kp; i += 256) SP(i, j) *= rs;
        __syncthreads();
        for (int i = j + 1 + ti; i <= kp; i += 16) {
            const double li = SP(i, j);
            const int cmax = (i < kp) ? i : kp - 1;
            for (int c = j + 1 + tj; c <= cmax; c += 16) SP(i, c) -= li * SP(c, j);
        }
        __syncthreads();
    }
    for (int i = ti; i < kp; i += 16)
        for (int j = tj; j < kp; j += 16)
            fr.lmat[(size_t)i * fr.ldl + j] = (j < i) ? SP(i, j) : ((j == i) ? diag[i] : 0.0);
    for (int j = tid; j < kp; j += 256) fr.yvec[j] = SP(kp, j);
    // inverse of every 16x16 diagonal block of L (one thread per column)
    if (tid < kp) {
        const int b = tid >> 4, jc = tid & 15, r0 = b * EKF_RB;
        double x[EKF_RB];
#pragma unroll
        for (int i = 0; i < EKF_RB; ++i) {
            double sacc = 0.0;
#pragma unroll
            for (int c = 0; c < EKF_RB; ++c)
                if (c >= jc && c < i) sacc += SP(r0 + i, r0 + c) * x[c];
            const double inv = 1.0 / diag[r0 + i];
            x[i] = (i < jc) ? 0.0 : ((i == jc) ? inv : -sacc * inv);
        }
#pragma unroll
        for (int i = 0; i < EKF_RB; ++i) {
            fr.dinv[(size_t)(r0 + i) * EKF_RB + jc] = x[i];
            fr.dop[(size_t)(b * 4 + (jc >> 2)) * 64 + i + 16 * (jc & 3)] = x[i];
        }
    }
    {   // -L blocks in MFMA A-operand order
        const int nb = kp / EKF_RB;
        const int total = nb * (nb - 1) / 2 * 256;
        for (int e = tid; e < total; e += 256) {
            const int lane = e & 63, r = (e >> 6) & 3, bq = e >> 8;
            int b = 1;
            while ((b + 1) * b / 2 <= bq) ++b;            // bq = b(b-1)/2 + q, q < b
            const int q = bq - b * (b - 1) / 2;
            fr.lop[e] = -SP(16 * b + (lane & 15), 16 * q + (lane >> 4) + 4 * r);
        }
    }
#undef SP
}

// --------------------------------------------------------------------------
// solve, fast path (kpad <= 128): blocked left-looking Cholesky, S dense in LDS.
//   phase 1  all waves : block column b -= L[:, <b] L[b, <b]^T   (v_mfma_f64_16x16x4)
//   phase 2  wave 0    : 16 pivots on up to 64 rows at once, lane = row, the 16
//                        panel entries of a row in registers, pivot row broadcast
//                        by v_readlane (no LDS, no barrier inside the chain)
//   phase 3  waves 1-3 : rows beyond wave 0's 64: in-lane triangular solve
// The residual rides along as row kp, so its factor row is y = L^-1 (z - h).
// --------------------------------------------------------------------------
typedef double sf64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double ekf_readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double ekf_rsqrt_f64(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-h * y, y, 0.5);
        y = __builtin_fma(y, e, y);
    }
    return y;
}

int ekf_solve_blocked_lds_bytes(int kpad, int k) {
    const size_t ldd = (size_t)kpad + 2;
    return (int)((((size_t)kpad + 1) * ldd + (size_t)k * (EKF_CAM + EKF_JCOLS) + kpad + 4) * sizeof(double));
}

// -L blocks of block column p in MFMA operand order (see EkfFrame::lop); one wave.
__device__ __forceinline__ void ekf_solve_emit_lop(const EkfFrame& fr, const double* S, int LD,
                                                   int nb, int p, int lane) {
    for (int i = p + 1; i < nb; ++i) {
        const size_t base = (size_t)(i * (i - 1) / 2 + p) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            fr.lop[base + r * 64 + lane] =
                -S[(EKF_RB * i + (lane & 15)) * LD + EKF_RB * p + (lane >> 4) + 4 * r];
    }
}

// Left-looking update of one 16x16 block:  S[rows of block i][cb .. cb+15] -= sum_{q in [q0,q1)} L_iq L_bq^T
// (b = cb / 16).  One wave; two independent MFMA chains.  Row block nb is the residual row (all 16
// operand rows alias row kp, only row 0 is stored).
__device__ __forceinline__ void ekf_solve_update_block(double* S, int LD, int kp, int i, int cb,
                                                       int q0, int q1, int c, int g) {
    const int rbase = EKF_RB * i;
    const double* arow = S + (size_t)min(rbase + c, kp) * LD + g;
    const double* brow = S + (size_t)(cb + c) * LD + g;
    sf64x4 t, t2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = S[min(rbase + g + 4 * r, kp) * LD + cb + c];
    int q = q0;
    for (; q + 1 < q1; q += 2) {
        double av[8], bv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            av[r] = arow[EKF_RB * q + 4 * r];     // A[i = c][k = g + 4r]
            bv[r] = brow[EKF_RB * q + 4 * r];     // B[k][j = c] = L[cb + c][k]
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[r], bv[r], t, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[4 + r], bv[4 + r], t2, 0, 0, 0);
        }
    }
    if (q < q1) {
        double av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { av[r] = arow[EKF_RB * q + 4 * r]; bv[r] = brow[EKF_RB * q + 4 * r]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) t = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[r], bv[r], t, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = rbase + g + 4 * r;
        if (row <= kp) S[row * LD + cb + c] = t[r] + t2[r];
    }
}

// MS = 5: m <= 32 detections, MS = 6: m <= 64.  Pair p -> (row r1 = p >> MS, detection
// jp = p & (2^MS - 1)); a thread's pairs all share jp (the block size is a multiple of 2^MS).
#define EKF_SOLVE_THREADS 512
template <int MS>
__global__ __launch_bounds__(EKF_SOLVE_THREADS) void ekf_solve_blocked_kernel(EkfFrame fr) {
    extern __shared__ __attribute__((aligned(16))) double b_sm[];
    constexpr int NT = EKF_SOLVE_THREADS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    const int k = fr.k, kp = fr.kpad, nb = kp / EKF_RB, LD = kp + 2, m = fr.m;
    double* S = b_sm;                                   // [kp + 1][LD], row kp = residual
    double* invd = S + (size_t)(kp + 1) * LD;           // [kp]  1 / L_jj
    double* ac = invd + kp + 2;                         // [k][10]  A[:, 0:10]
    double* hl = ac + (size_t)k * EKF_CAM;              // [k][13]
    int nstamp = 0;
#define EKF_STAMP() do { if (fr.stamps && tid == 0) fr.stamps[nstamp] = clock64(); ++nstamp; } while (0)
    EKF_STAMP();
    // ---- build S = A[:, support] H^T + R (lower triangle).
    // asup (row stride LD) is copied flat into S with coalesced 16-byte loads; entry
    // S[r1][3jp+d] is then computed in place from S[r1][3jp..3jp+2] = asup[r1][3jp..3jp+2].
    constexpr int RSTEP = NT >> MS;                     // rows advanced per pair slot
    constexpr int NPAIR = 128 / RSTEP;                  // pair slots: 128 rows per batch
    const int jp = tid & ((1 << MS) - 1);
    const bool jvalid = jp < m;
    {
        const int n16 = (k * LD) >> 1;                  // k * LD is even (LD even)
        const double2* src = reinterpret_cast<const double2*>(fr.asup);
        double2* dst = reinterpret_cast<double2*>(S);
        constexpr int NV = 10;                          // 10 * 512 double2 = 10240 doubles >= 96 * 98
        double2 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + NT * i;
            v[i] = src[min(e, n16 - 1)];
        }
        double acv[4], hv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + NT * i, r = min(e >> 4, k - 1), cc = e & 15;
            acv[i] = fr.amat[(int64_t)r * fr.lda + min(cc, EKF_CAM - 1)];
            hv[i] = fr.jac[min(e, k * EKF_JLD - 1)];    // jac is [k][16], contiguous
        }
        const double rres = fr.resid[min(tid, k - 1)];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + NT * i;
            if (e < n16) dst[e] = v[i];
        }
        for (int e = tid + NT * NV; e < n16; e += NT) dst[e] = src[e];   // (k * LD > 10240 doubles)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + NT * i, r = e >> 4, cc = e & 15;
            if (r < k && cc < EKF_CAM) ac[r * EKF_CAM + cc] = acv[i];
            if (r < k && cc < EKF_JCOLS) hl[r * EKF_JCOLS + cc] = hv[i];
        }
        // padding rows k..kp-1 = identity, residual row kp
        const int tr = tid >> 4, tc = tid & 15;
        for (int r = k + tr; r <= kp; r += NT / 16) {
            const int hi = (r < kp) ? (r | 15) : kp - 1;
            for (int cc = tc; cc <= hi; cc += 16) S[r * LD + cc] = (r < kp && cc == r) ? 1.0 : 0.0;
        }
        __syncthreads();
        EKF_STAMP();
        if (tid < k) S[kp * LD + tid] = rres;
    }
    {
        double hreg[3][EKF_JCOLS];                      // this thread's 3 Jacobian rows
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int a = 0; a < EKF_JCOLS; ++a)
                hreg[d][a] = hl[(3 * min(jp, m - 1) + d) * EKF_JCOLS + a];
        for (int r0 = 0; r0 < k; r0 += 128) {           // one batch for k <= 128
#pragma unroll
            for (int u = 0; u < NPAIR; ++u) {
                const int r1 = r0 + (tid >> MS) + RSTEP * u;
                if (jvalid && r1 < k && r1 >= 3 * jp) {
                    double* sp = S + r1 * LD + 3 * jp;
                    const double a0 = sp[0], a1 = sp[1], a2 = sp[2];
                    double a10[EKF_CAM];
#pragma unroll
                    for (int bb = 0; bb < EKF_CAM; ++bb) a10[bb] = ac[r1 * EKF_CAM + bb];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const int r2 = 3 * jp + d;
                        double acc = (r1 == r2) ? fr.nz.r_unc : 0.0;
#pragma unroll
                        for (int bb = 0; bb < EKF_CAM; ++bb) acc += a10[bb] * hreg[d][bb];
                        acc += a0 * hreg[d][10] + a1 * hreg[d][11] + a2 * hreg[d][12];
                        sp[d] = (r2 <= r1) ? acc : 0.0;   // strict upper part of a diagonal block = 0
                    }
                }
            }
        }
    }
    __syncthreads();
    EKF_STAMP();

    // ---- blocked left-looking Cholesky.  8 waves:
    //  (A) all waves : block column b -= L[:, b-1] L[b, b-1]^T  (the older terms were already applied)
    //  (B) waves 0-2 : the 16-pivot chain of block b (lane = row, v_readlane broadcast)
    //      waves 3-6 : meanwhile apply the finished block columns 0..b-1 to block column b+1
    //      wave  7   : meanwhile emit the panel kernel's operands of block column b-1
    int bad = 0;
    for (int b = 0; b < nb; ++b) {
        const int cb = EKF_RB * b;
        if (b > 0) {
            for (int i = b + wave; i <= nb; i += 8) ekf_solve_update_block(S, LD, kp, i, cb, b - 1, b, c, g);
            __syncthreads();
        }
        EKF_STAMP();
        //  wave 0 : rows cb .. cb+63
        //  wave 1 : lanes 0-15 the diagonal rows again (every wave needs the pivot rows in its own
        //           lanes), lanes 16-31 the rows of I_16 -- the same recurrence turns e_i into row i
        //           of L_bb^-T, i.e. column i of Dinv_b, for free --, lanes 32-63 rows cb+64 .. cb+95
        //  wave 2 : lanes 0-15 diagonal rows, lanes 16-63 rows cb+96 .. cb+143 (kp <= 128: enough)
        const int xrow0 = (wave == 1) ? cb + 64 : cb + 96;
        if (wave <= 1 || (wave == 2 && xrow0 <= kp)) {
            int row;
            if (wave == 0 || lane < EKF_RB) row = cb + lane;
            else if (wave == 1) row = (lane < 32) ? -1 : xrow0 + (lane - 32);
            else row = xrow0 + (lane - EKF_RB);
            const bool ident = row < 0;
            const bool store = !ident && (wave == 0 || lane >= EKF_RB) && row <= kp;
            const double* src = S + (size_t)min(max(row, 0), kp) * LD + cb;
            double a[EKF_RB], ys[EKF_RB];
#pragma unroll
            for (int x = 0; x < EKF_RB; ++x) a[x] = ident ? ((x == lane - EKF_RB) ? 1.0 : 0.0) : src[x];
#pragma unroll
            for (int j = 0; j < EKF_RB; ++j) {
                const double d = ekf_readlane_f64(a[j], j);
                bad |= !(d > 0.0);
                const double y = ekf_rsqrt_f64(d);
                ys[j] = y;
                const double lj = a[j] * y;
                a[j] = lj;
#pragma unroll
                for (int x = j + 1; x < EKF_RB; ++x) {
                    const double lx = ekf_readlane_f64(lj, x);
                    a[x] = __builtin_fma(-lj, lx, a[x]);
                }
            }
            if (store) {
                double* dst = S + (size_t)row * LD + cb;
#pragma unroll
                for (int x = 0; x < EKF_RB; ++x) dst[x] = a[x];
            }
            if (ident) {                                 // a[x] = Dinv_b[x][i], i = lane - 16
                const int i = lane - EKF_RB;
#pragma unroll
                for (int x = 0; x < EKF_RB; ++x) {
                    fr.dinv[(size_t)(cb + x) * EKF_RB + i] = a[x];
                    fr.dop[(size_t)(b * 4 + (i >> 2)) * 64 + x + 16 * (i & 3)] = a[x];
                }
            }
            if (tid == 0) {
#pragma unroll
                for (int j = 0; j < EKF_RB; ++j) invd[cb + j] = ys[j];
            }
        } else if (wave >= 3 && wave <= 6) {
            if (b > 0 && b + 1 < nb)                     // rows of blocks b+1 .. nb (nb = residual row)
                for (int i = b + 1 + (wave - 3); i <= nb; i += 4)
                    ekf_solve_update_block(S, LD, kp, i, cb + EKF_RB, 0, b, c, g);
        } else if (wave == 7 && b > 0) {
            ekf_solve_emit_lop(fr, S, LD, nb, b - 1, lane);
        }
        __syncthreads();
        EKF_STAMP();
    }
    if (bad && lane == 0) atomicOr(fr.status, 1);
    // (block column nb-1 has no -L blocks below it)
    for (int j = tid; j < kp; j += NT) fr.yvec[j] = S[kp * LD + j];
    if (fr.wdbg) {                                      // dense L for tests only
        for (int i = tid >> 4; i < kp; i += NT / 16)
            for (int j = tid & 15; j < kp; j += 16)
                fr.lmat[(size_t)i * fr.ldl + j] = (j <= i) ? S[i * LD + j] : 0.0;
    }
    EKF_STAMP();
#undef EKF_STAMP
}

// --------------------------------------------------------------------------
// solve, column-streaming blocked Cholesky (kpad <= 192).  16 waves, one workgroup.
//   * only 4 block columns of S live in LDS (ring); a finished block column b is emitted from
//     the pivot waves' registers in MFMA-operand order (-L, `lop`) -- the layout the panel kernel
//     reads anyway -- and is read back from there by the later left-looking updates;
//   * per block column b:   (A) all waves : column b -= L[:, b-1] L[b, b-1]^T        (LDS operands)
//                           (B) pivot waves: 16-pivot chain (lane = row, v_readlane broadcast),
//                               one more wave carries the diagonal rows + I_16 -> Dinv_b;
//                               every other wave meanwhile: column b+1 -= (q = b-1 term) and
//                               column b+2 is BUILT (S = A_supp H^T + R from asupt, ac, hl) and gets
//                               its q <= b-1 terms (lop for q <= b-2, LDS for q = b-1);
//     so the S build and all but one update term per column hide behind the pivot chain.
//   * the residual rides along as row kp (its factor row is y = L^-1 (z - h)).
// --------------------------------------------------------------------------
#define SV_T 512
#define SV_CLD 18

__device__ __forceinline__ size_t sv_lop_index(int i, int q) { return (size_t)(i * (i - 1) / 2 + q) * 256; }

// C-in / C-out of row block i of the column buffer `tgt` (block column tc), one wave
struct SvAcc { sf64x4 t, t2; };
__device__ __forceinline__ void sv_acc_load(SvAcc& a, const double* tgt, int kp, int i, int c, int g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a.t[r] = tgt[min(EKF_RB * i + g + 4 * r, kp) * SV_CLD + c];
        a.t2[r] = 0.0;
    }
}
__device__ __forceinline__ void sv_acc_store(const SvAcc& a, double* tgt, int kp, int i, int c, int g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = EKF_RB * i + g + 4 * r;
        if (row <= kp) tgt[row * SV_CLD + c] = a.t[r] + a.t2[r];
    }
}
// one q term, operands from the LDS buffer of block column q:  -= L_iq L_tq^T
__device__ __forceinline__ void sv_term_lds(SvAcc& a, const double* qbuf, int kp, int i, int tc, int c, int g,
                                            bool second = false) {
    const double* ar = qbuf + (size_t)min(EKF_RB * i + c, kp) * SV_CLD + g;   // residual block: rows alias kp
    const double* br = qbuf + (size_t)(EKF_RB * tc + c) * SV_CLD + g;
    double av[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { av[r] = ar[4 * r]; bv[r] = br[4 * r]; }
    if (second) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[r], bv[r], a.t2, 0, 0, 0);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.t = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[r], bv[r], a.t, 0, 0, 0);
    }
}
// q in [0, q1), operands from global: lop (-L blocks) and, for the residual block, y
__device__ __forceinline__ void sv_terms_glb(SvAcc& a, const double* __restrict__ lop,
                                             const double* __restrict__ yv, int nb, int i, int tc, int q1,
                                             int g, int lane) {
    if (q1 <= 0) return;
    // operands of term q+1 are in flight while the MFMAs of term q run
    double av[4], bv[4], an[4], bn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        av[r] = (i < nb) ? lop[sv_lop_index(i, 0) + r * 64 + lane] : -yv[g + 4 * r];
        bv[r] = -lop[sv_lop_index(tc, 0) + r * 64 + lane];
    }
    for (int q = 0; q < q1; ++q) {
        const int qn = min(q + 1, q1 - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            an[r] = (i < nb) ? lop[sv_lop_index(i, qn) + r * 64 + lane] : -yv[EKF_RB * qn + g + 4 * r];
            bn[r] = -lop[sv_lop_index(tc, qn) + r * 64 + lane];
        }
        if (q & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a.t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv[r], a.t2, 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) a.t = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv[r], a.t, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { av[r] = an[r]; bv[r] = bn[r]; }
    }
}

// S entries of row block i of block column tc (rows 16i.., columns 16tc..): one wave, lane =
// (row rr = lane & 15, column group cg = lane >> 4 -> 4 columns)
struct SvBuild { double as[4][3]; double rs[4]; };
// global loads of one block (branch-free, clamped) ...
__device__ __forceinline__ void sv_build_load(const EkfFrame& fr, SvBuild& w, int kp, int nb, int i, int tc,
                                              int lane) {
    const int k = fr.k, rr = lane & 15, cg = lane >> 4;
    const int r1c = min((i == nb) ? kp : EKF_RB * i + rr, k - 1);
    const int r2b = EKF_RB * tc + 4 * cg;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int r2c = min(r2b + x, k - 1);
        const int jp = (r2c * 43691) >> 17;            // r2 / 3 for r2 < 98304
        const double* ap = fr.asupt + (size_t)(3 * jp) * fr.ldast + r1c;
        w.as[x][0] = ap[0];
        w.as[x][1] = ap[fr.ldast];
        w.as[x][2] = ap[2 * fr.ldast];
        w.rs[x] = fr.resid[r2c];
    }
}
// ... and the arithmetic: S entries of row block i of block column tc (rows 16i.., columns 16tc..);
// one wave, lane = (row rr = lane & 15, column group cg = lane >> 4 -> 4 columns)
__device__ __forceinline__ void sv_build_finish(const EkfFrame& fr, const SvBuild& w, double* tgt,
                                                const double* ac, const double* hl, int kp, int nb, int i,
                                                int tc, int lane) {
    const int k = fr.k, rr = lane & 15, cg = lane >> 4;
    const bool is_res = (i == nb);
    if (is_res && rr != 0) return;                     // residual row block: one row
    const int r1 = is_res ? kp : EKF_RB * i + rr;
    const int r1c = min(r1, k - 1);
    const int r2b = EKF_RB * tc + 4 * cg;
    double a10[EKF_CAM];
#pragma unroll
    for (int bb = 0; bb < EKF_CAM; ++bb) a10[bb] = ac[r1c * EKF_CAM + bb];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int r2 = r2b + x;
        const double* h2 = hl + min(r2, k - 1) * EKF_JCOLS;
        double acc = (r1 == r2) ? fr.nz.r_unc : 0.0;
#pragma unroll
        for (int bb = 0; bb < EKF_CAM; ++bb) acc += a10[bb] * h2[bb];
        acc += w.as[x][0] * h2[10] + w.as[x][1] * h2[11] + w.as[x][2] * h2[12];
        const bool pad = (r1 >= k) || (r2 >= k);       // identity padding rows / columns
        double v = (r2 > r1) ? 0.0 : acc;              // strict upper part of a diagonal block
        v = pad ? ((r1 == r2) ? 1.0 : 0.0) : v;
        v = is_res ? ((r2 < k) ? w.rs[x] : 0.0) : v;
        tgt[r1 * SV_CLD + 4 * cg + x] = v;
    }
}
__device__ __forceinline__ void sv_build_block(const EkfFrame& fr, double* tgt, const double* ac,
                                               const double* hl, int kp, int nb, int i, int tc, int lane) {
    SvBuild w;
    sv_build_load(fr, w, kp, nb, i, tc, lane);
    sv_build_finish(fr, w, tgt, ac, hl, kp, nb, i, tc, lane);
}

// number of block columns resident in LDS: all of them when they fit (kpad <= 112), else a ring
int ekf_solve_stream_ring(int kpad, int k) {
    const size_t col = (size_t)(kpad + 1) * SV_CLD * sizeof(double);
    const size_t fixed = ((size_t)k * (EKF_CAM + EKF_JCOLS) + 8) * sizeof(double);
    int rs = (int)((160 * 1024 - fixed) / col);
    const int nb = kpad / EKF_RB;
    if (rs > nb) rs = nb;
    return rs < 4 ? 4 : rs;
}
int ekf_solve_stream_lds_bytes(int kpad, int k) {
    return (int)(((size_t)ekf_solve_stream_ring(kpad, k) * (kpad + 1) * SV_CLD +
                  (size_t)k * (EKF_CAM + EKF_JCOLS) + 8) * sizeof(double));
}

__global__ __launch_bounds__(SV_T) void ekf_solve_stream_kernel(EkfFrame fr, int RS) {
    extern __shared__ __attribute__((aligned(16))) double v_sm[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int k = fr.k, kp = fr.kpad, nb = kp / EKF_RB, rows = kp + 1;
    // RS block columns live in LDS: all of them (RS == nb) or a ring (column c in slot c % RS,
    // overwritten when column c + RS is built, i.e. during iteration c + RS - 2)
    double* ring = v_sm;                                         // [RS][rows][SV_CLD]
    double* ac = ring + (size_t)RS * rows * SV_CLD;              // [k][10]
    const bool all_resident = RS >= nb;
    double* hl = ac + (size_t)k * EKF_CAM;                       // [k][13]
    auto colbuf = [&](int col) { return ring + (size_t)(col % RS) * rows * SV_CLD; };
    constexpr int NW = SV_T / 64;
    int nstamp = 0;
#define EKF_STAMP() do { if (fr.stamps && tid == 0) fr.stamps[nstamp] = clock64(); ++nstamp; } while (0)
    EKF_STAMP();
    // prologue.  Block columns built before the first pivot chain: all of them when they are all
    // resident, else 0 and 1 (column c >= 2 is then built during iteration c - 2).  The global
    // loads of this wave's first PB blocks are issued together with the ac / hl staging loads:
    // one exposed memory round trip instead of one per block.
    constexpr int PB = 5;
    const int ncol0 = all_resident ? nb : (nb > 1 ? 2 : 1);
    int ntot = 0;
    for (int tc = 0; tc < ncol0; ++tc) ntot += nb - tc + 1;
    SvBuild pw[PB];
    int pi[PB], ptc[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        int u = wave + NW * j, tc = 0;
        const bool has = u < ntot;
        u = has ? u : 0;
        while (u >= nb - tc + 1) { u -= nb - tc + 1; ++tc; }
        pi[j] = has ? tc + u : -1;
        ptc[j] = tc;
        if (has) sv_build_load(fr, pw[j], kp, nb, pi[j], tc, lane);
    }
    EKF_STAMP();
    for (int e = tid; e < k * 16; e += SV_T) {
        const int r = e >> 4, cc = e & 15;
        if (cc < EKF_CAM) ac[r * EKF_CAM + cc] = fr.amat[(int64_t)r * fr.lda + cc];
        if (cc < EKF_JCOLS) hl[r * EKF_JCOLS + cc] = fr.jac[e];
    }
    EKF_STAMP();
    __syncthreads();
    EKF_STAMP();
#pragma unroll
    for (int j = 0; j < PB; ++j)
        if (pi[j] >= 0) sv_build_finish(fr, pw[j], colbuf(ptc[j]), ac, hl, kp, nb, pi[j], ptc[j], lane);
    for (int u0 = wave + NW * PB; u0 < ntot; u0 += NW) {      // (more than PB blocks per wave: rare)
        int u = u0, tc = 0;
        while (u >= nb - tc + 1) { u -= nb - tc + 1; ++tc; }
        sv_build_block(fr, colbuf(tc), ac, hl, kp, nb, tc + u, tc, lane);
    }
    __syncthreads();
    EKF_STAMP();

    int bad = 0;
    for (int b = 0; b < nb; ++b) {
        const int cb = EKF_RB * b;
        double* cur = colbuf(b);
        if (b > 0) {   // (A) the one term that could not be applied earlier
            for (int i = b + wave; i <= nb; i += NW) {
                SvAcc a;
                sv_acc_load(a, cur, kp, i, c, g);
                sv_term_lds(a, colbuf(b - 1), kp, i, b, c, g);
                sv_acc_store(a, cur, kp, i, c, g);
            }
            __syncthreads();
        }
        EKF_STAMP();
        // (B) roles.  rows of this block column: cb .. kp (kp = residual row)
        const int nrows = kp - cb + 1;
        const int nrw = (nrows > 64) ? (nrows - 64 + 47) / 48 : 0;   // extra row waves after wave 0
        if (wave <= nrw + 1) {
            // wave 0: rows cb..cb+63.  waves 1..nrw: lanes 0-15 the diagonal rows again (every wave
            // needs the pivot rows in its own lanes), lanes 16-63 rows cb+64+48(w-1)...  wave nrw+1:
            // diagonal rows + the rows of I_16: e_i -> row i of L_bb^-T = column i of Dinv_b.
            const bool idw = (wave == nrw + 1);
            int row;
            if (wave == 0 || lane < EKF_RB) row = cb + lane;
            else if (idw) row = -1;
            else row = cb + 64 + 48 * (wave - 1) + (lane - EKF_RB);
            const bool ident = idw && lane >= EKF_RB && lane < 2 * EKF_RB;
            const bool live = !idw && (wave == 0 || lane >= EKF_RB) && row <= kp;
            const double* src = cur + (size_t)min(max(row, 0), kp) * SV_CLD;
            double a[EKF_RB];
#pragma unroll
            for (int x = 0; x < EKF_RB; ++x) a[x] = (row < 0) ? ((x == lane - EKF_RB) ? 1.0 : 0.0) : src[x];
#pragma unroll
            for (int j = 0; j < EKF_RB; ++j) {
                const double d = ekf_readlane_f64(a[j], j);
                bad |= !(d > 0.0);
                const double y = ekf_rsqrt_f64(d);
                const double lj = a[j] * y;
                a[j] = lj;
#pragma unroll
                for (int x = j + 1; x < EKF_RB; ++x) {
                    const double lx = ekf_readlane_f64(lj, x);
                    a[x] = __builtin_fma(-lj, lx, a[x]);
                }
            }
            if (live) {
                double* dst = cur + (size_t)row * SV_CLD;
#pragma unroll
                for (int x = 0; x < EKF_RB; ++x) dst[x] = a[x];
                if (row == kp) {                         // residual row: y of this block
#pragma unroll
                    for (int x = 0; x < EKF_RB; ++x) fr.yvec[cb + x] = a[x];
                } else if (row >= cb + EKF_RB) {         // -L block (row/16, b) in MFMA operand order
                    const size_t base = sv_lop_index(row >> 4, b) + (row & 15);
#pragma unroll
                    for (int x = 0; x < EKF_RB; ++x) fr.lop[base + (x >> 2) * 64 + 16 * (x & 3)] = -a[x];
                }
                if (fr.wdbg && row < kp) {               // dense L for tests only
#pragma unroll
                    for (int x = 0; x < EKF_RB; ++x) fr.lmat[(size_t)row * fr.ldl + cb + x] = (cb + x <= row) ? a[x] : 0.0;
                }
            }
            if (ident) {                                 // a[x] = Dinv_b[x][i], i = lane - 16
                const int i = lane - EKF_RB;
#pragma unroll
                for (int x = 0; x < EKF_RB; ++x) {
                    fr.dinv[(size_t)(cb + x) * EKF_RB + i] = a[x];
                    fr.dop[(size_t)(b * 4 + (i >> 2)) * 64 + x + 16 * (i & 3)] = a[x];
                }
            }
        } else {
            const int fw = wave - (nrw + 2), nfw = NW - (nrw + 2);   // free waves
            // block column b+1: the q = b-1 term (operands still in LDS)
            if (b >= 1 && b + 1 < nb)
                for (int i = b + 1 + fw; i <= nb; i += nfw) {
                    SvAcc a;
                    sv_acc_load(a, colbuf(b + 1), kp, i, c, g);
                    sv_term_lds(a, colbuf(b - 1), kp, i, b + 1, c, g);
                    sv_acc_store(a, colbuf(b + 1), kp, i, c, g);
                }
            // block column b+2: built here unless it already is; then every finished term q <= b-1
            // (from LDS while column q is still resident, else from the emitted -L blocks)
            const int tc = b + 2;
            if (tc < nb)
                for (int i = tc + fw; i <= nb; i += nfw) {
                    if (!all_resident) sv_build_block(fr, colbuf(tc), ac, hl, kp, nb, i, tc, lane);
                    if (b >= 1) {
                        SvAcc a;
                        sv_acc_load(a, colbuf(tc), kp, i, c, g);
                        int q = 0;
                        for (; q < b && !(q + RS >= nb || b <= q + RS - 3); ++q) {}   // first resident q
                        sv_terms_glb(a, fr.lop, fr.yvec, nb, i, tc, q, g, lane);
                        for (; q < b; ++q) sv_term_lds(a, colbuf(q), kp, i, tc, c, g, (q & 1) != 0);
                        sv_acc_store(a, colbuf(tc), kp, i, c, g);
                    }
                }
        }
        __syncthreads();
        EKF_STAMP();
    }
    if (bad && lane == 0) atomicOr(fr.status, 1);
    EKF_STAMP();
#undef EKF_STAMP
}

void ekf_launch_solve(const EkfFrame& fr, hipStream_t s) {
    static bool once = false;
    if (!once) {   // > 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ekf_solve_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ekf_solve_blocked_kernel<5>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ekf_solve_blocked_kernel<6>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    static const bool old_path = getenv("EKF_SOLVE_OLD") != nullptr;
    if (!old_path) {
        static bool once2 = false;
        if (!once2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ekf_solve_stream_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            once2 = true;
        }
        hipLaunchKernelGGL(ekf_solve_stream_kernel, dim3(1), dim3(SV_T),
                           ekf_solve_stream_lds_bytes(fr.kpad, fr.k), s, fr,
                           ekf_solve_stream_ring(fr.kpad, fr.k));
        return;
    }
    const int lds = ekf_solve_blocked_lds_bytes(fr.kpad, fr.k);
    if (fr.kpad <= 128 && fr.m <= 32)
        hipLaunchKernelGGL(ekf_solve_blocked_kernel<5>, dim3(1), dim3(EKF_SOLVE_THREADS), lds, s, fr);
    else if (fr.kpad <= 128)
        hipLaunchKernelGGL(ekf_solve_blocked_kernel<6>, dim3(1), dim3(EKF_SOLVE_THREADS), lds, s, fr);
    else
        hipLaunchKernelGGL(ekf_solve_kernel, dim3(1), dim3(256), ekf_solve_lds_bytes(fr.kpad), s, fr);
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
            part += v * fr.yvec[row];
            wp[(int64_t)row * fr.ldw + col0 + j] = (T)v;
            if (fr.wdbg) fr.wdbg[(int64_t)row * fr.ldw + col0 + j] = v;
        }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);                    // dx[col0 + j] in every lane group
    const int col = col0 + j;
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
        case 1: return ekf_panel_mfma_go<T, 1>(fr, s);
        case 2: return ekf_panel_mfma_go<T, 2>(fr, s);
        case 3: return ekf_panel_mfma_go<T, 3>(fr, s);
        case 4: return ekf_panel_mfma_go<T, 4>(fr, s);
        case 5: return ekf_panel_mfma_go<T, 5>(fr, s);
        case 6: return ekf_panel_mfma_go<T, 6>(fr, s);
        case 7: return ekf_panel_mfma_go<T, 7>(fr, s);
        case 8: return ekf_panel_mfma_go<T, 8>(fr, s);
        case 9: return ekf_panel_mfma_go<T, 9>(fr, s);
        case 10: return ekf_panel_mfma_go<T, 10>(fr, s);
        case 11: return ekf_panel_mfma_go<T, 11>(fr, s);
        default: return ekf_panel_mfma_go<T, 12>(fr, s);
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
