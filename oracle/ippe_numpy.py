"""CPU restatement of the detection -> pose step (TEST INFRASTRUCTURE ONLY: imported by tests/ and
__graft_entry__.smoke(), never by the product package).

Reference call site: ``BaseFilter.estimate_pose_of_markers`` (/root/reference/filters/base_filter.py:92-171) calls
``cv2.solvePnP(marker_points, corners, K, dist, flags=cv2.SOLVEPNP_IPPE_SQUARE)`` once per marker.  OpenCV is a pip
dependency of the reference (``opencv-contrib-python``, unpinned in its requirements) and is neither vendored in
/root/reference nor installed here, and the reference holds no recorded corner / pose pairs: **parity unpinned**.
This file restates the published algorithm behind the flag (Collins & Bartoli, "Infinitesimal Plane-based Pose
Estimation", IJCV 2014) and OpenCV's documented camera model (pinhole + Brown-Conrady ``k1 k2 p1 p2 k3 [k4 k5 k6]``;
``undistortPoints`` = 5 fixed-point iterations), written independently of the HIP kernel where that is cheap: the
homography by a linear solve (the kernel uses a closed form), the largest singular value and the least-squares
translation through ``numpy.linalg``, the rotation vector through SciPy.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def object_points(marker_size: float) -> np.ndarray:
    """base_filter.py:113-121 (the order SOLVEPNP_IPPE_SQUARE prescribes)."""
    h = marker_size / 2.0
    return np.array([[-h, h, 0.0], [h, h, 0.0], [h, -h, 0.0], [-h, -h, 0.0]])


def _dist8(dist) -> np.ndarray:
    d = np.zeros(8)
    v = np.asarray([] if dist is None else dist, dtype=np.float64).reshape(-1)
    d[: v.size] = v
    return d


def project_points(points_cam: np.ndarray, camera_matrix, dist=None) -> np.ndarray:
    """Forward camera model (cv2.projectPoints for points already in the camera frame): [n,3] -> pixels [n,2]."""
    k = _dist8(dist)
    fx, fy, cx, cy = camera_matrix[0][0], camera_matrix[1][1], camera_matrix[0][2], camera_matrix[1][2]
    x = points_cam[:, 0] / points_cam[:, 2]
    y = points_cam[:, 1] / points_cam[:, 2]
    r2 = x * x + y * y
    cd = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2)
    xd = x * cd + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
    yd = y * cd + k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
    return np.stack([fx * xd + cx, fy * yd + cy], axis=1)


def undistort_points(pixels: np.ndarray, camera_matrix, dist=None, iterations: int = 5) -> np.ndarray:
    """cv2.undistortPoints without a new camera matrix: pixels [n,2] -> normalised image points [n,2]."""
    k = _dist8(dist)
    fx, fy, cx, cy = camera_matrix[0][0], camera_matrix[1][1], camera_matrix[0][2], camera_matrix[1][2]
    x0 = (pixels[:, 0] - cx) / fx
    y0 = (pixels[:, 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iterations):
        r2 = x * x + y * y
        icd = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x = (x0 - dx) * icd
        y = (y0 - dy) * icd
    return np.stack([x, y], axis=1)


def _homography(obj_xy: np.ndarray, img: np.ndarray) -> np.ndarray:
    """Exact homography through four correspondences, h22 = 1 (8 x 8 linear solve)."""
    a, b = [], []
    for (X, Y), (x, y) in zip(obj_xy, img):
        a.append([X, Y, 1, 0, 0, 0, -x * X, -x * Y]); b.append(x)
        a.append([0, 0, 0, X, Y, 1, -y * X, -y * Y]); b.append(y)
    h = np.linalg.solve(np.array(a), np.array(b))
    return np.append(h, 1.0).reshape(3, 3)


def _translation(rot: np.ndarray, obj: np.ndarray, img: np.ndarray):
    pr = obj @ rot.T
    a, b = [], []
    for (X, Y, Z), (x, y) in zip(pr, img):
        a.append([1, 0, -x]); b.append(x * Z - X)
        a.append([0, 1, -y]); b.append(y * Z - Y)
    t = np.linalg.lstsq(np.array(a), np.array(b), rcond=None)[0]
    pc = pr + t
    err = float(np.sum((pc[:, :2] / pc[:, 2:3] - img) ** 2))
    return t, err


def ippe_square(corners_px: np.ndarray, marker_size: float, camera_matrix, dist=None):
    """One marker: pixel corners [4,2] -> (tvec, rvec) of the solution with the smaller reprojection error, and
    both candidate solutions [(tvec, rvec, err), ...] (best first)."""
    obj = object_points(marker_size)
    img = undistort_points(np.asarray(corners_px, dtype=np.float64).reshape(4, 2), camera_matrix, dist)
    hm = _homography(obj[:, :2], img)
    p, q = hm[0, 2], hm[1, 2]
    jac = np.array([[hm[0, 0] - hm[2, 0] * p, hm[0, 1] - hm[2, 1] * p],
                    [hm[1, 0] - hm[2, 0] * q, hm[1, 1] - hm[2, 1] * q]])
    # rotation that takes the optical axis to the ray through (p, q, 1)
    d = np.array([p, q, 1.0]) / np.sqrt(p * p + q * q + 1.0)
    w = np.cross([0.0, 0.0, 1.0], d)
    wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    rv = np.eye(3) + wx + wx @ wx / (1.0 + d[2])
    bm = (np.array([[1, 0, -p], [0, 1, -q]]) @ rv)[:, :2]
    am = np.linalg.solve(bm, jac)
    gamma = np.linalg.svd(am, compute_uv=False)[0]
    r22 = am / gamma
    b0 = np.sqrt(max(1.0 - r22[:, 0] @ r22[:, 0], 0.0))
    b1 = np.sqrt(max(1.0 - r22[:, 1] @ r22[:, 1], 0.0))
    if r22[:, 0] @ r22[:, 1] > 0:
        b1 = -b1
    sols = []
    for sgn in (1.0, -1.0):
        c0 = np.array([r22[0, 0], r22[1, 0], sgn * b0])
        c1 = np.array([r22[0, 1], r22[1, 1], sgn * b1])
        rot = rv @ np.stack([c0, c1, np.cross(c0, c1)], axis=1)
        t, err = _translation(rot, obj, img)
        sols.append((t, Rotation.from_matrix(rot).as_rotvec(), err))
    if sols[1][2] < sols[0][2]:
        sols.reverse()
    return sols[0][0], sols[0][1], sols


def estimate_pose_of_markers(corners, marker_size: float, camera_matrix, dist=None) -> np.ndarray:
    """base_filter.py:92-171: [m,6] = [tvec | rvec] for every marker."""
    c = np.asarray(corners, dtype=np.float64).reshape(-1, 4, 2)
    out = np.zeros((c.shape[0], 6))
    for j in range(c.shape[0]):
        t, r, _ = ippe_square(c[j], marker_size, camera_matrix, dist)
        out[j, :3], out[j, 3:] = t, r
    return out
