// Detection -> pose front end, batched: the pose of every detected square marker from its four image corners
// (reference: BaseFilter.estimate_pose_of_markers, filters/base_filter.py:92-171 -- one cv2.solvePnP(...,
// flags=SOLVEPNP_IPPE_SQUARE) per marker in a Python loop).  One thread per marker, f64 throughout.
//
// OpenCV itself is not in the reference tree (a pip dependency) and not in this image; what follows restates the
// PUBLISHED algorithm behind that flag -- Collins & Bartoli, "Infinitesimal Plane-based Pose Estimation", IJCV 2014,
// specialised to a square -- in the order OpenCV documents it:
//   1. pixel corners -> normalised, undistorted image points (pinhole + Brown-Conrady k1 k2 p1 p2 k3 [k4 k5 k6],
//      the fixed-point iteration of cv::undistortPoints, 5 iterations);
//   2. the homography H from the marker plane (corners (-s/2, s/2), (s/2, s/2), (s/2, -s/2), (-s/2, -s/2), the order
//      IPPE_SQUARE prescribes and base_filter.py:113-121 passes) to those points, exact for four points
//      (square -> quadrilateral in closed form);
//   3. IPPE: from the Jacobian J of H at the marker centre and the centre's image v, the rotation R_v that takes the
//      optical axis to the ray through v, the 2x2 factor A = B^-1 J with B = [I | -v] R_v, its largest singular
//      value gamma, and the two rotations whose upper-left 2x2 block is A / gamma;
//   4. for either rotation the translation by linear least squares over the four corners;
//   5. the solution with the smaller reprojection error (normalised image plane) is returned as [tvec | rvec],
//      rvec = axis * angle of the rotation.
// No fixture of the reference pins these numbers ("parity unpinned"): tests compare the kernel with a NumPy
// restatement of the same steps (oracle/ippe_numpy.py) and with the poses the corners were projected from.
// One KNOWN structural difference to the reference's call: base_filter.py passes float32 corners and object points, and
// cv::undistortPoints returns CV_32F for float32 input, so OpenCV's normalised points are rounded to f32 before IPPE;
// this kernel stays in f64 throughout.  Expect ~1e-7 relative differences in tvec / rvec against a cv2 run (the
// as-written EKF amplifies such differences over long free runs: SURVEY F5).
#include "ekf_kernels.h"

namespace {

struct Vec3 { double x, y, z; };

__device__ __forceinline__ void ippe_undistort(const EkfCamera& cam, double u, double v, double& x, double& y) {
    const double x0 = (u - cam.cx) / cam.fx, y0 = (v - cam.cy) / cam.fy;
    x = x0;
    y = y0;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const double r2 = x * x + y * y;
        const double icd = (1.0 + ((cam.k[7] * r2 + cam.k[6]) * r2 + cam.k[5]) * r2) /
                           (1.0 + ((cam.k[4] * r2 + cam.k[1]) * r2 + cam.k[0]) * r2);
        const double dx = 2.0 * cam.k[2] * x * y + cam.k[3] * (r2 + 2.0 * x * x);
        const double dy = cam.k[2] * (r2 + 2.0 * y * y) + 2.0 * cam.k[3] * x * y;
        x = (x0 - dx) * icd;
        y = (y0 - dy) * icd;
    }
}

// translation for a given rotation: minimise sum_i |(X_i' + t_x, Y_i' + t_y) - (Z_i' + t_z) p_i|^2, P_i' = R P_i
// (normal equations of the 8 x 3 system [1 0 -x_i; 0 1 -y_i] t = [x_i Z' - X'; y_i Z' - Y']), and the reprojection
// error of the resulting pose
__device__ __forceinline__ double ippe_translation(const double R[3][3], const double px[4], const double py[4],
                                                   double h, Vec3& t) {
    const double ox[4] = {-h, h, h, -h}, oy[4] = {h, h, -h, -h};
    double sxx = 0.0, syy = 0.0, sx = 0.0, sy = 0.0, bx = 0.0, by = 0.0, bz = 0.0;
    double Xr[4], Yr[4], Zr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        Xr[i] = R[0][0] * ox[i] + R[0][1] * oy[i];
        Yr[i] = R[1][0] * ox[i] + R[1][1] * oy[i];
        Zr[i] = R[2][0] * ox[i] + R[2][1] * oy[i];
        const double rx = px[i] * Zr[i] - Xr[i], ry = py[i] * Zr[i] - Yr[i];
        sx += px[i];
        sy += py[i];
        sxx += px[i] * px[i];
        syy += py[i] * py[i];
        bx += rx;
        by += ry;
        bz += -px[i] * rx - py[i] * ry;
    }
    // M = [[4, 0, -sx], [0, 4, -sy], [-sx, -sy, sxx + syy]] (symmetric positive definite): eliminate t_x, t_y
    const double m22 = (sxx + syy) - 0.25 * (sx * sx + sy * sy);
    t.z = (bz + 0.25 * (sx * bx + sy * by)) / m22;
    t.x = 0.25 * (bx + sx * t.z);
    t.y = 0.25 * (by + sy * t.z);
    double err = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double zc = Zr[i] + t.z, ex = (Xr[i] + t.x) / zc - px[i], ey = (Yr[i] + t.y) / zc - py[i];
        err += ex * ex + ey * ey;
    }
    return err;
}

// axis * angle of a rotation matrix (the inverse of Rodrigues' formula)
__device__ __forceinline__ Vec3 ippe_rotvec(const double R[3][3]) {
    const double rx = R[2][1] - R[1][2], ry = R[0][2] - R[2][0], rz = R[1][0] - R[0][1];
    const double s = 0.5 * sqrt(rx * rx + ry * ry + rz * rz);
    double c = 0.5 * (R[0][0] + R[1][1] + R[2][2] - 1.0);
    c = fmin(1.0, fmax(-1.0, c));
    const double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0.0) return Vec3{0.0, 0.0, 0.0};
        // angle ~ pi: the axis from the symmetric part, R + I = 2 a a^T
        double ax = sqrt(fmax(0.5 * (R[0][0] + 1.0), 0.0));
        double ay = sqrt(fmax(0.5 * (R[1][1] + 1.0), 0.0)) * (R[0][1] < 0.0 ? -1.0 : 1.0);
        double az = sqrt(fmax(0.5 * (R[2][2] + 1.0), 0.0)) * (R[0][2] < 0.0 ? -1.0 : 1.0);
        if (fabs(ax) < fabs(ay) && fabs(ax) < fabs(az) && ((R[1][2] > 0.0) != (ay * az > 0.0))) az = -az;
        const double f = theta / sqrt(ax * ax + ay * ay + az * az);
        return Vec3{ax * f, ay * f, az * f};
    }
    const double f = 0.5 * theta / s;
    return Vec3{rx * f, ry * f, rz * f};
}

__global__ __launch_bounds__(64) void ekf_ippe_square_kernel(const double* __restrict__ corners, int count, double half,
                                                             EkfCamera cam, double* __restrict__ out) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= count) return;
    // 1. normalised image points of the corners, in the order (-h, h), (h, h), (h, -h), (-h, -h)
    double px[4], py[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ippe_undistort(cam, corners[8 * j + 2 * i], corners[8 * j + 2 * i + 1], px[i], py[i]);
    // 2. homography: unit square (0,0), (1,0), (1,1), (0,1) <-> corners 3, 2, 1, 0, composed with
    //    (X, Y) -> ((X + h) / 2h, (Y + h) / 2h)
    double H[3][3];
    {
        const double x0 = px[3], y0 = py[3], x1 = px[2], y1 = py[2], x2 = px[1], y2 = py[1], x3 = px[0], y3 = py[0];
        const double dx1 = x1 - x2, dx2 = x3 - x2, sxs = x0 - x1 + x2 - x3;
        const double dy1 = y1 - y2, dy2 = y3 - y2, sys = y0 - y1 + y2 - y3;
        const double den = dx1 * dy2 - dx2 * dy1;
        const double g = (sxs * dy2 - dx2 * sys) / den, hh = (dx1 * sys - sxs * dy1) / den;
        const double a = x1 - x0 + g * x1, b = x3 - x0 + hh * x3, d = y1 - y0 + g * y1, e = y3 - y0 + hh * y3;
        const double s = 0.5 / half;                         // columns scaled by 1 / 2h, third = a/2 + b/2 + c ...
        const double h22 = 0.5 * g + 0.5 * hh + 1.0;
        H[0][0] = a * s / h22;
        H[0][1] = b * s / h22;
        H[0][2] = (0.5 * a + 0.5 * b + x0) / h22;
        H[1][0] = d * s / h22;
        H[1][1] = e * s / h22;
        H[1][2] = (0.5 * d + 0.5 * e + y0) / h22;
        H[2][0] = g * s / h22;
        H[2][1] = hh * s / h22;
        H[2][2] = 1.0;
    }
    // 3. IPPE
    const double p = H[0][2], q = H[1][2];
    const double j00 = H[0][0] - H[2][0] * p, j01 = H[0][1] - H[2][1] * p;
    const double j10 = H[1][0] - H[2][0] * q, j11 = H[1][1] - H[2][1] * q;
    double Rv[3][3];
    {
        const double t = sqrt(p * p + q * q + 1.0), a = p / t, b = q / t, c = 1.0 / t, k = 1.0 / (1.0 + c);
        Rv[0][0] = 1.0 - a * a * k;  Rv[0][1] = -a * b * k;       Rv[0][2] = a;
        Rv[1][0] = -a * b * k;       Rv[1][1] = 1.0 - b * b * k;  Rv[1][2] = b;
        Rv[2][0] = -a;               Rv[2][1] = -b;               Rv[2][2] = c;
    }
    // B = [I | -v] Rv (its first two columns), A = B^-1 J
    const double b00 = Rv[0][0] - p * Rv[2][0], b01 = Rv[0][1] - p * Rv[2][1];
    const double b10 = Rv[1][0] - q * Rv[2][0], b11 = Rv[1][1] - q * Rv[2][1];
    const double idet = 1.0 / (b00 * b11 - b01 * b10);
    const double a00 = idet * (b11 * j00 - b01 * j10), a01 = idet * (b11 * j01 - b01 * j11);
    const double a10 = idet * (-b10 * j00 + b00 * j10), a11 = idet * (-b10 * j01 + b00 * j11);
    const double ata00 = a00 * a00 + a10 * a10, ata01 = a00 * a01 + a10 * a11, ata11 = a01 * a01 + a11 * a11;
    const double gamma = sqrt(0.5 * (ata00 + ata11 + sqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01)));
    const double r00 = a00 / gamma, r01 = a01 / gamma, r10 = a10 / gamma, r11 = a11 / gamma;
    // third row of the first two columns: b0^2 = 1 - |c0|^2, b1^2 = 1 - |c1|^2, b0 b1 = -c0.c1
    double b0 = sqrt(fmax(1.0 - r00 * r00 - r10 * r10, 0.0));
    double b1 = sqrt(fmax(1.0 - r01 * r01 - r11 * r11, 0.0));
    if (r00 * r01 + r10 * r11 > 0.0) b1 = -b1;
    double best_err = 0.0;
    Vec3 best_t{0.0, 0.0, 0.0}, best_r{0.0, 0.0, 0.0};
#pragma unroll
    for (int sol = 0; sol < 2; ++sol) {
        const double s0 = sol ? -b0 : b0, s1 = sol ? -b1 : b1;
        // R~ = [c0 c1 c0 x c1], R = Rv R~
        const double c0[3] = {r00, r10, s0}, c1[3] = {r01, r11, s1};
        const double c2[3] = {c0[1] * c1[2] - c0[2] * c1[1], c0[2] * c1[0] - c0[0] * c1[2], c0[0] * c1[1] - c0[1] * c1[0]};
        double R[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            R[r][0] = Rv[r][0] * c0[0] + Rv[r][1] * c0[1] + Rv[r][2] * c0[2];
            R[r][1] = Rv[r][0] * c1[0] + Rv[r][1] * c1[1] + Rv[r][2] * c1[2];
            R[r][2] = Rv[r][0] * c2[0] + Rv[r][1] * c2[1] + Rv[r][2] * c2[2];
        }
        Vec3 t;
        const double err = ippe_translation(R, px, py, half, t);
        if (sol == 0 || err < best_err) {
            best_err = err;
            best_t = t;
            best_r = ippe_rotvec(R);
        }
    }
    out[6 * j + 0] = best_t.x;
    out[6 * j + 1] = best_t.y;
    out[6 * j + 2] = best_t.z;
    out[6 * j + 3] = best_r.x;
    out[6 * j + 4] = best_r.y;
    out[6 * j + 5] = best_r.z;
}

}  // namespace

void ekf_launch_ippe_square(const double* corners_dev, int count, double marker_size, const EkfCamera& cam,
                            double* poses_dev, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(ekf_ippe_square_kernel, dim3((count + 63) / 64), dim3(64), 0, s, corners_dev, count,
                       0.5 * marker_size, cam, poses_dev);
}
