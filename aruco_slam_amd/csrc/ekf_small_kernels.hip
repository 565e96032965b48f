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

// --------------------------------------------------------------------------
// gather
// --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ekf_gather_kernel(EkfFrame fr) {
    extern __shared__ __attribute__((aligned(16))) double g_sm[];
    double* hs = g_sm;                                  // [k][13]
    int* lmc = reinterpret_cast<int*>(g_sm + fr.k * EKF_JCOLS);
    const int tid = threadIdx.x;
    const int m = fr.m;
    if (tid < m) {
        const int c0 = EKF_CAM + EKF_LM * fr.idx[tid];
        double cam[EKF_CAM], lm[3], h[3], J[3][EKF_JCOLS];
        for (int a = 0; a < EKF_CAM; ++a) cam[a] = fr.state[a];
        for (int d = 0; d < 3; ++d) lm[d] = fr.state[c0 + d];
        ekf_measure(cam, lm, h, J);
        for (int d = 0; d < 3; ++d)
            for (int a = 0; a < EKF_JCOLS; ++a) hs[(3 * tid + d) * EKF_JCOLS + a] = J[d][a];
        lmc[tid] = c0;
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
    const int c = blockIdx.x * 64 + (tid & 63);   // < ncols <= ld by construction
    const int g = tid >> 6;                        // wave index: wave-uniform landmark
    const T* __restrict__ P = static_cast<const T*>(fr.cov);
    const int64_t ld = fr.ld;
    double pc[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) {
        pc[a] = (double)P[a * ld + c];
        if (a == c) pc[a] += ekf_qdiag(a, fr.dims, fr.nz);
    }
    for (int j = g; j < m; j += 4) {
        const int c0 = lmc[j];
        double pl[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            pl[d] = (double)P[(int64_t)(c0 + d) * ld + c];
            if (c0 + d == c) pl[d] += fr.nz.q_lm;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double* hr = hs + (3 * j + d) * EKF_JCOLS;
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < EKF_CAM; ++a) acc += hr[a] * pc[a];
#pragma unroll
            for (int e = 0; e < 3; ++e) acc += hr[10 + e] * pl[e];
            fr.amat[(int64_t)(3 * j + d) * fr.lda + c] = acc;
        }
    }
    for (int r = fr.k + g; r < fr.kpad; r += 4) fr.amat[(int64_t)r * fr.lda + c] = 0.0;
}

template <typename T>
void ekf_launch_gather(const EkfFrame& fr, hipStream_t s) {
    const size_t lds = (size_t)fr.k * EKF_JCOLS * sizeof(double) + (size_t)fr.m * sizeof(int) + 16;
    hipLaunchKernelGGL(ekf_gather_kernel<T>, dim3(fr.ncols / 64), dim3(256), lds, s, fr);
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
    return (int)((((size_t)kpad + 1) * ldd + (size_t)k * EKF_JCOLS + (size_t)k * EKF_CAM + kpad + 2) *
                 sizeof(double));
}

__global__ __launch_bounds__(256) void ekf_solve_blocked_kernel(EkfFrame fr) {
    extern __shared__ __attribute__((aligned(16))) double b_sm[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    const int k = fr.k, kp = fr.kpad, nb = kp / EKF_RB, LD = kp + 2, m = fr.m;
    double* S = b_sm;                                   // [kp + 1][LD], row kp = residual
    double* invd = S + (size_t)(kp + 1) * LD;           // [kp]  1 / L_jj
    double* hl = invd + kp + (kp & 1);                  // [k][13]
    double* ac = hl + (size_t)k * EKF_JCOLS;            // [k][10]  A[:, 0:10]

    for (int e = tid; e < k * EKF_JCOLS; e += 256)
        hl[e] = fr.jac[(size_t)(e / EKF_JCOLS) * EKF_JLD + (e % EKF_JCOLS)];
    for (int e = tid; e < k * EKF_CAM; e += 256)
        ac[e] = fr.amat[(int64_t)(e / EKF_CAM) * fr.lda + (e % EKF_CAM)];
    for (int e = tid; e < (kp + 1) * LD; e += 256) S[e] = 0.0;
    __syncthreads();
    // S = A[:, support] H^T + R (lower); pairs (row r1, detection jp) -> 3 entries
    for (int p = tid; p < k * m; p += 256) {
        const int r1 = p / m, jp = p - r1 * m;
        if (3 * jp > r1) continue;
        const int c0 = fr.lmcol[jp];
        const double* ar = fr.amat + (int64_t)r1 * fr.lda + c0;
        const double al0 = ar[0], al1 = ar[1], al2 = ar[2];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int r2 = 3 * jp + d;
            if (r2 <= r1) {
                const double* h2 = hl + r2 * EKF_JCOLS;
                double acc = (r1 == r2) ? fr.nz.r_unc : 0.0;
#pragma unroll
                for (int bb = 0; bb < EKF_CAM; ++bb) acc += ac[r1 * EKF_CAM + bb] * h2[bb];
                acc += al0 * h2[10] + al1 * h2[11] + al2 * h2[12];
                S[r1 * LD + r2] = acc;
            }
        }
    }
    for (int r = k + tid; r < kp; r += 256) S[r * LD + r] = 1.0;       // identity padding
    for (int j = tid; j < k; j += 256) S[kp * LD + j] = fr.resid[j];    // augmented row
    __syncthreads();

    int bad = 0;
    for (int b = 0; b < nb; ++b) {
        const int cb = EKF_RB * b;
        if (b > 0) {
            // ---- phase 1: row blocks b .. nb (nb = residual row), round-robin over waves
            for (int i = b + wave; i <= nb; i += 4) {
                const int rbase = EKF_RB * i;
                const int arow = min(rbase + c, kp);              // residual block: all rows alias kp
                sf64x4 t;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = S[min(rbase + g + 4 * r, kp) * LD + cb + c];
                for (int q = 0; q < b; ++q) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kc = EKF_RB * q + g + 4 * r;
                        const double av = -S[arow * LD + kc];     // A[i = c][k = g + 4r]
                        const double bv = S[(cb + c) * LD + kc];  // B[k][j = c] = L[cb + c][k]
                        t = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, t, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rbase + g + 4 * r;
                    if (row <= kp) S[row * LD + cb + c] = t[r];
                }
            }
            __syncthreads();
        }
        // ---- phase 2: wave 0, lane = row cb + lane
        const int last_w0 = min(cb + 63, kp);
        if (wave == 0) {
            const int row = cb + lane;
            const bool valid = row <= kp;
            const double* src = S + (size_t)min(row, kp) * LD + cb;
            double a[EKF_RB];
#pragma unroll
            for (int x = 0; x < EKF_RB; ++x) a[x] = src[x];
#pragma unroll
            for (int j = 0; j < EKF_RB; ++j) {
                const double d = ekf_readlane_f64(a[j], j);
                bad |= !(d > 0.0);
                const double y = ekf_rsqrt_f64(d);
                const double lj = a[j] * y;
                a[j] = lj;
                if (lane == 0) invd[cb + j] = y;
#pragma unroll
                for (int x = j + 1; x < EKF_RB; ++x) {
                    const double lx = ekf_readlane_f64(lj, x);
                    a[x] = __builtin_fma(-lj, lx, a[x]);
                }
            }
            if (valid) {
                double* dst = S + (size_t)row * LD + cb;
#pragma unroll
                for (int x = 0; x < EKF_RB; ++x) dst[x] = a[x];
            }
        }
        __syncthreads();
        // ---- phase 3: rows below wave 0's window
        if (last_w0 < kp) {
            const int row = cb + 64 + (tid - 64);
            if (wave > 0 && row <= kp) {
                double* rp = S + (size_t)row * LD + cb;
                double x[EKF_RB];
#pragma unroll
                for (int j = 0; j < EKF_RB; ++j) {
                    double acc = rp[j];
#pragma unroll
                    for (int q = 0; q < j; ++q) acc = __builtin_fma(-x[q], S[(cb + j) * LD + cb + q], acc);
                    x[j] = acc * invd[cb + j];
                }
#pragma unroll
                for (int j = 0; j < EKF_RB; ++j) rp[j] = x[j];
            }
            __syncthreads();
        }
    }
    if (bad && lane == 0) atomicOr(fr.status, 1);

    // ---- outputs: y, dense L (debug / tests), -L in MFMA operand order, Dinv
    for (int j = tid; j < kp; j += 256) fr.yvec[j] = S[kp * LD + j];
    for (int e = tid; e < kp * kp; e += 256) {
        const int i = e / kp, j = e - i * kp;
        fr.lmat[(size_t)i * fr.ldl + j] = (j <= i) ? S[i * LD + j] : 0.0;
    }
    {
        const int total = nb * (nb - 1) / 2 * 256;
        for (int e = tid; e < total; e += 256) {
            const int ln = e & 63, r = (e >> 6) & 3, bq = e >> 8;
            int b = 1;
            while ((b + 1) * b / 2 <= bq) ++b;
            const int q = bq - b * (b - 1) / 2;
            fr.lop[e] = -S[(16 * b + (ln & 15)) * LD + 16 * q + (ln >> 4) + 4 * r];
        }
    }
    if (tid < kp) {
        const int b = tid >> 4, jc = tid & 15, r0 = b * EKF_RB;
        double x[EKF_RB];
#pragma unroll
        for (int i = 0; i < EKF_RB; ++i) {
            double sacc = 0.0;
#pragma unroll
            for (int cc = 0; cc < EKF_RB; ++cc)
                if (cc >= jc && cc < i) sacc += S[(r0 + i) * LD + r0 + cc] * x[cc];
            const double inv = invd[r0 + i];
            x[i] = (i < jc) ? 0.0 : ((i == jc) ? inv : -sacc * inv);
        }
#pragma unroll
        for (int i = 0; i < EKF_RB; ++i) {
            fr.dinv[(size_t)(r0 + i) * EKF_RB + jc] = x[i];
            fr.dop[(size_t)(b * 4 + (jc >> 2)) * 64 + i + 16 * (jc & 3)] = x[i];
        }
    }
}

void ekf_launch_solve(const EkfFrame& fr, hipStream_t s) {
    static bool once = false;
    if (!once) {   // > 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ekf_solve_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ekf_solve_blocked_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    if (fr.kpad <= 128)
        hipLaunchKernelGGL(ekf_solve_blocked_kernel, dim3(1), dim3(256),
                           ekf_solve_blocked_lds_bytes(fr.kpad, fr.k), s, fr);
    else
        hipLaunchKernelGGL(ekf_solve_kernel, dim3(1), dim3(256), ekf_solve_lds_bytes(fr.kpad), s, fr);
}

// --------------------------------------------------------------------------
// panel: 32 columns per workgroup, blocked forward substitution with the
// inverted diagonal blocks:  W_b = Dinv_b (A_b - sum_{q<b} L_bq W_q).
// --------------------------------------------------------------------------
int ekf_panel_lds_bytes(int kpad) {
    return (int)(((size_t)kpad * 32 + (size_t)EKF_RB * kpad + 256 + 512 + 256 + 64) * sizeof(double));
}

template <typename T>
__global__ __launch_bounds__(256) void ekf_panel_kernel(EkfFrame fr) {
    extern __shared__ __attribute__((aligned(16))) double p_sm[];
    const int kp = fr.kpad;
    double* wl = p_sm;                   // [kp][32]   A chunk, overwritten by W
    double* lrow = wl + (size_t)kp * 32; // [16][kp]   current block row of L
    double* dv = lrow + (size_t)EKF_RB * kp;  // [16][16]
    double* tb = dv + 256;               // [16][32]
    double* red = tb + 512;              // [8][32]
    double* misc = red + 256;            // [64]
    const int tid = threadIdx.x, c = tid & 31, g = tid >> 5;
    const int col0 = blockIdx.x * 32;
    for (int r = g; r < kp; r += 8) wl[r * 32 + c] = fr.amat[(int64_t)r * fr.lda + col0 + c];
    const int nb = kp / EKF_RB;
    for (int b = 0; b < nb; ++b) {
        const int r0 = b * EKF_RB;
        for (int e = tid; e < EKF_RB * r0; e += 256) {
            const int rr = e / r0, q = e - rr * r0;
            lrow[rr * kp + q] = fr.lmat[(size_t)(r0 + rr) * fr.ldl + q];
        }
        dv[tid] = fr.dinv[(size_t)b * 256 + tid];
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int rr = g + 8 * hh;
            double t = wl[(r0 + rr) * 32 + c];
            for (int q = 0; q < r0; ++q) t -= lrow[rr * kp + q] * wl[q * 32 + c];
            tb[rr * 32 + c] = t;
        }
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int rr = g + 8 * hh;
            double w = 0.0;
            for (int i = 0; i <= rr; ++i) w += dv[rr * EKF_RB + i] * tb[i * 32 + c];
            wl[(r0 + rr) * 32 + c] = w;
        }
        __syncthreads();
    }
    // write W (cov dtype, k-major) and dx = W^T y
    T* __restrict__ wp = static_cast<T*>(fr.wpanel);
    double part = 0.0;
    for (int r = g; r < kp; r += 8) {
        const double w = wl[r * 32 + c];
        part += w * fr.yvec[r];
        wp[(int64_t)r * fr.ldw + col0 + c] = (T)w;
        if (fr.wdbg) fr.wdbg[(int64_t)r * fr.ldw + col0 + c] = w;
    }
    red[g * 32 + c] = part;
    __syncthreads();
    if (g == 0) {
        double dx = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) dx += red[q * 32 + c];
        const int col = col0 + c;
        // extended_kalman_filter.py:134-135: xyz and every landmark move,
        // delta[3:7] is dropped, delta[7:10] feeds the quaternion
        if (col < 3 || (col >= EKF_CAM && col < fr.dims)) {
            const double nv = fr.state[col] + dx;
            fr.state[col] = nv;
            if (col < 3) misc[col] = nv;
        } else if (col >= 7 && col < EKF_CAM) {
            misc[col] = dx;
        }
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        if (tid == 0) {
            double q[4] = {fr.state[3], fr.state[4], fr.state[5], fr.state[6]};
            const double err[3] = {misc[7], misc[8], misc[9]};
            ekf_quat_inject(q, err, fr.quat_mode);
            for (int i = 0; i < 4; ++i) fr.state[3 + i] = q[i];
            for (int i = 0; i < 3; ++i) fr.state[7 + i] = 0.0;   // :152
            if (fr.traj_row) {
                for (int i = 0; i < 3; ++i) fr.traj_row[i] = misc[i];
                for (int i = 0; i < 4; ++i) fr.traj_row[3 + i] = q[i];
            }
        }
    }
}

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
void ekf_launch_panel(const EkfFrame& fr, int variant, hipStream_t s) {
    if (variant != 1) {
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
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ekf_panel_kernel<T>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    hipLaunchKernelGGL(ekf_panel_kernel<T>, dim3(fr.ncols / 32), dim3(256),
                       ekf_panel_lds_bytes(fr.kpad), s, fr);
}
template void ekf_launch_panel<float>(const EkfFrame&, int, hipStream_t);
template void ekf_launch_panel<double>(const EkfFrame&, int, hipStream_t);

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
