// Device-side helpers shared by the EKF kernels (gfx950 only).
//
// Measurement model: closed form of the SymPy-generated lambdas of the
// reference (extended_kalman_filter.py:292-353), evaluated at e = 0 which
// holds at every call site (:152 resets the error state).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EKF_CAM 10
#define EKF_LM 3
#define EKF_JCOLS 13   // [c(3) q(4) e(3) l(3)], extended_kalman_filter.py:327-343
#define EKF_JLD 16     // padded row length of a stored Jacobian row
#define EKF_RB 16      // row-block size of the small factorisation / panel solve

struct EkfNoise {
    double q_cam, q_err, q_lm, r_unc;
};

// process-noise diagonal of state dimension i, extended_kalman_filter.py:98-104
__device__ __forceinline__ double ekf_qdiag(int i, int dims, const EkfNoise& nz) {
    if (i < 3) return nz.q_cam;
    if (i < 7) return 0.0;
    if (i < 10) return nz.q_err;
    return i < dims ? nz.q_lm : 0.0;
}

// h (3) and dh (3 x 13) for one landmark.  cam = state[0:10], lm = state[c0:c0+3].
__device__ inline void ekf_measure(const double* __restrict__ cam, const double* __restrict__ lm,
                                   double h[3], double J[3][EKF_JCOLS]) {
    const double a = cam[3];
    const double u[3] = {cam[4], cam[5], cam[6]};
    const double v[3] = {lm[0] - cam[0], lm[1] - cam[1], lm[2] - cam[2]};
    const double uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
    const double uv = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
    const double s = a * a + uu;
    const double is = 1.0 / s;
    const double cx[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2],
                          u[0] * v[1] - u[1] * v[0]};
    const double d0 = a * a - uu;
    for (int i = 0; i < 3; ++i) h[i] = (d0 * v[i] + 2.0 * uv * u[i] - 2.0 * a * cx[i]) * is;
    // skew(w)[i][j]
    const double su[3][3] = {{0.0, -u[2], u[1]}, {u[2], 0.0, -u[0]}, {-u[1], u[0], 0.0}};
    const double sv[3][3] = {{0.0, -v[2], v[1]}, {v[2], 0.0, -v[0]}, {-v[1], v[0], 0.0}};
    double dq[3][4];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            const double dij = (i == j) ? 1.0 : 0.0;
            const double rt = (d0 * dij + 2.0 * u[i] * u[j] - 2.0 * a * su[i][j]) * is;  // R(q)^T
            J[i][j] = -rt;
            J[i][10 + j] = rt;
            const double g = -2.0 * v[i] * u[j] + 2.0 * u[i] * v[j] + 2.0 * uv * dij +
                             2.0 * a * sv[i][j];
            dq[i][1 + j] = g * is - 2.0 * h[i] * u[j] * is;
        }
        const double g0 = 2.0 * a * v[i] - 2.0 * cx[i];
        dq[i][0] = g0 * is - 2.0 * h[i] * a * is;
    }
    for (int i = 0; i < 3; ++i) {
        for (int c = 0; c < 4; ++c) J[i][3 + c] = dq[i][c];
        for (int j = 0; j < 3; ++j) {
            // E = [-u^T ; a I - skew(u)]  (4 x 3)
            double acc = -dq[i][0] * u[j];
            for (int r = 0; r < 3; ++r)
                acc += dq[i][1 + r] * (((r == j) ? a : 0.0) - su[r][j]);
            J[i][7 + j] = acc;
        }
    }
}

// Camera quaternion injection, extended_kalman_filter.py:138-149.
// mode 0: as written (both arrays read scalar-LAST by SciPy), mode 1: scalar-first.
__device__ inline void ekf_quat_inject(double q[4], const double err[3], int mode) {
    double d[4] = {1.0, 0.5 * err[0], 0.5 * err[1], 0.5 * err[2]};
    const double nq = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double nd = 1.0 / sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
    double x1, y1, z1, w1, x2, y2, z2, w2;
    if (mode == 0) {
        x1 = d[0] * nd; y1 = d[1] * nd; z1 = d[2] * nd; w1 = d[3] * nd;
        x2 = q[0] * nq; y2 = q[1] * nq; z2 = q[2] * nq; w2 = q[3] * nq;
    } else {
        w1 = d[0] * nd; x1 = d[1] * nd; y1 = d[2] * nd; z1 = d[3] * nd;
        w2 = q[0] * nq; x2 = q[1] * nq; y2 = q[2] * nq; z2 = q[3] * nq;
    }
    const double w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
    const double x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    const double y = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
    const double z = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
    const double nr = 1.0 / sqrt(w * w + x * x + y * y + z * z);
    q[0] = w * nr; q[1] = x * nr; q[2] = y * nr; q[3] = z * nr;
}
