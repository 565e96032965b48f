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
#define EKF_JLD 20     // row length of a stored Jacobian row (13 used by EKF, 20 by EKF_Rotations)
#define EKF_RB 16      // row-block size of the small factorisation / panel solve

// Filter models.  0: reference EKF (extended_kalman_filter.py): landmark = xyz, 3 rows per
// detection, 3 x 13 Jacobian.  1: reference EKF_Rotations (ekf_with_rotations.py): landmark =
// [xyz | qw qx qy qz | e(3)], 7 rows per detection [xyz_cl ; q_cl], 7 x 20 Jacobian.
template <int MODEL> struct EkfModel;
template <> struct EkfModel<0> { static constexpr int RD = 3, LMD = 3, JC = 13, NDET16 = 7; };
template <> struct EkfModel<1> { static constexpr int RD = 7, LMD = 10, JC = 20, NDET16 = 4; };

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
// No fused-multiply-add contraction in the measurement model: it is evaluated by several kernels /
// workgroup roles, and every instance has to produce the same bits whatever it is inlined into.
__device__ inline void ekf_measure(const double* __restrict__ cam, const double* __restrict__ lm,
                                   double h[3], double J[3][EKF_JCOLS]) {
#pragma clang fp contract(off)
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
// (in two halves: the norm of the old quaternion does not depend on the frame's result -- the front kernel takes it
// before the substitution, off the tail of the launch; same expressions, same bits)
__device__ inline double ekf_quat_rnorm(const double q[4]) {
#pragma clang fp contract(off)
    return 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
}
__device__ inline void ekf_quat_inject_n(double q[4], double nq, const double err[3], int mode) {
#pragma clang fp contract(off)      // same bits wherever it is inlined (front kernel, panel kernel, injection kernel)
    double d[4] = {1.0, 0.5 * err[0], 0.5 * err[1], 0.5 * err[2]};
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
__device__ inline void ekf_quat_inject(double q[4], const double err[3], int mode) {
    ekf_quat_inject_n(q, ekf_quat_rnorm(q), err, mode);
}


// EKF_Rotations measurement: h = [R(q_c)^-1 (l - c) ; q_c^-1 (x) q_l] and its 7 x 20 Jacobian in the
// column order [c(3) qc(4) ec(3) | l(3) ql(4) el(3)] (ekf_with_rotations.py:363-422), at ec = el = 0
// (the camera error state is reset every update, the landmarks' error states are never written).
// sympy's Quaternion.inverse() is conj / |q|^2.
__device__ inline void ekf_measure_rot(const double* __restrict__ cam, const double* __restrict__ lm,
                                       double h[7], double J[7][20]) {
#pragma clang fp contract(off)
    double h3[3], j13[3][EKF_JCOLS];
    ekf_measure(cam, lm, h3, j13);
    for (int i = 0; i < 3; ++i) {
        h[i] = h3[i];
        for (int cc = 0; cc < 10; ++cc) J[i][cc] = j13[i][cc];
        for (int cc = 0; cc < 3; ++cc) J[i][10 + cc] = j13[i][10 + cc];
        for (int cc = 13; cc < 20; ++cc) J[i][cc] = 0.0;
    }
    const double p[4] = {cam[3], cam[4], cam[5], cam[6]};
    const double r[4] = {lm[3], lm[4], lm[5], lm[6]};
    const double is = 1.0 / (p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
    const double cj[4] = {p[0], -p[1], -p[2], -p[3]};
    // q_cl = conj(p) (x) r / |p|^2
    double q[4];
    q[0] = (cj[0] * r[0] - cj[1] * r[1] - cj[2] * r[2] - cj[3] * r[3]) * is;
    q[1] = (cj[0] * r[1] + r[0] * cj[1] + cj[2] * r[3] - cj[3] * r[2]) * is;
    q[2] = (cj[0] * r[2] + r[0] * cj[2] + cj[3] * r[1] - cj[1] * r[3]) * is;
    q[3] = (cj[0] * r[3] + r[0] * cj[3] + cj[1] * r[2] - cj[2] * r[1]) * is;
    for (int i = 0; i < 4; ++i) h[3 + i] = q[i];
    // a (x) b = R(b) a ;  a (x) b = L(a) b
    const double rm[4][4] = {{r[0], -r[1], -r[2], -r[3]}, {r[1], r[0], r[3], -r[2]},
                             {r[2], -r[3], r[0], r[1]}, {r[3], r[2], -r[1], r[0]}};
    const double lmx[4][4] = {{cj[0], -cj[1], -cj[2], -cj[3]}, {cj[1], cj[0], -cj[3], cj[2]},
                              {cj[2], cj[3], cj[0], -cj[1]}, {cj[3], -cj[2], cj[1], cj[0]}};
    double dp[4][4], dr[4][4];
    for (int i = 0; i < 4; ++i)
        for (int cc = 0; cc < 4; ++cc) {
            dp[i][cc] = rm[i][cc] * ((cc == 0) ? 1.0 : -1.0) * is - 2.0 * q[i] * p[cc] * is;
            dr[i][cc] = lmx[i][cc] * is;
        }
    // E(x) = d((1, e) (x) x)/de at 0: row 0 = -u^T, rows 1..3 = a I - skew(u)
    auto emap = [](const double* x, double e[4][3]) {
        const double a = x[0], u0 = x[1], u1 = x[2], u2 = x[3];
        e[0][0] = -u0; e[0][1] = -u1; e[0][2] = -u2;
        e[1][0] = a;   e[1][1] = u2;  e[1][2] = -u1;
        e[2][0] = -u2; e[2][1] = a;   e[2][2] = u0;
        e[3][0] = u1;  e[3][1] = -u0; e[3][2] = a;
    };
    double ep[4][3], er[4][3];
    emap(p, ep);
    emap(r, er);
    for (int i = 0; i < 4; ++i) {
        for (int cc = 0; cc < 3; ++cc) { J[3 + i][cc] = 0.0; J[3 + i][10 + cc] = 0.0; }
        for (int cc = 0; cc < 4; ++cc) { J[3 + i][3 + cc] = dp[i][cc]; J[3 + i][13 + cc] = dr[i][cc]; }
        for (int jx = 0; jx < 3; ++jx) {
            double a1 = 0.0, a2 = 0.0;
            for (int cc = 0; cc < 4; ++cc) { a1 += dp[i][cc] * ep[cc][jx]; a2 += dr[i][cc] * er[cc][jx]; }
            J[3 + i][7 + jx] = a1;
            J[3 + i][17 + jx] = a2;
        }
    }
}

template <int MODEL>
__device__ __forceinline__ void ekf_measure_model(const double* cam, const double* lm, double* h,
                                                  double (*J)[EkfModel<MODEL>::JC]);
template <>
__device__ __forceinline__ void ekf_measure_model<0>(const double* cam, const double* lm, double* h,
                                                     double (*J)[13]) { ekf_measure(cam, lm, h, J); }
template <>
__device__ __forceinline__ void ekf_measure_model<1>(const double* cam, const double* lm, double* h,
                                                     double (*J)[20]) { ekf_measure_rot(cam, lm, h, J); }
