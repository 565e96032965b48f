"""CPU oracle for the EKF-SLAM update path  --  TEST INFRASTRUCTURE ONLY.

This is a NumPy/SciPy restatement of the reference algorithm
(/root/reference/filters/extended_kalman_filter.py:58-290).  It exists so that
the HIP path can be checked against something that runs anywhere.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; the product package ``aruco_slam_amd`` never does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference in the build container and stores its inputs/outputs as fixtures
under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file
against every one of them (h/dh lambdas, teacher-forced steps, free-run,
large-n checksums, the quaternion rule).

Two arithmetic modes, same results up to rounding:

* ``mode="reference_ops"``  -- the reference's own operation sequence
  (dense N x N process-noise matrix, dense->CSR of the whole covariance,
  ``spsolve(S, I)``, gain ``P H^T S^-1``, dense ``(I - K H) @ P``).  This is
  the timed CPU baseline.
* ``mode="fast"``  -- the algorithmically necessary work only
  (rank-k downdate ``P - (P H^T) S^-1 (H P)`` through a Cholesky factor).

State layout (extended_kalman_filter.py:29-34,46-51):
``[x y z | qw qx qy qz | ex ey ez | l0x l0y l0z | l1x ...]``, covariance dense
row-major ``N x N``, ``N = 3 n + 10``.
"""
from __future__ import annotations

import numpy as np

# noise constants: extended_kalman_filter.py:21-27
INITIAL_CAMERA_UNCERTAINTY = 0.1
INITIAL_LANDMARK_UNCERTAINTY = 0.7
R_UNCERTAINTY = 0.9
Q_UNCERTAINTY_CAM = 0.3
Q_ERROR_UNCERTAINTY_CAM = 0.5
Q_UNCERTAINTY_LM = 0.01

CAM = 10
LM = 3


# --------------------------------------------------------------------------
# measurement model, closed form of the SymPy lambdas
# (extended_kalman_filter.py:292-353; variable order :327-343)
# --------------------------------------------------------------------------
def _skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def h_closed(x13):
    """h(x) = R(dq*q)^-1 (l - c) with dq = (1, e), evaluated at e = 0.

    ``to_rotation_matrix()`` divides by |q|^2 (extended_kalman_filter.py:317),
    so h = f / s with s = |q|^2 also for non-unit q.
    """
    x13 = np.asarray(x13, dtype=np.float64)
    c, q, lm = x13[0:3], x13[3:7], x13[10:13]
    a, u = q[0], q[1:4]
    v = lm - c
    s = a * a + u @ u
    f = (a * a - u @ u) * v + 2.0 * (u @ v) * u - 2.0 * a * np.cross(u, v)
    return f / s


def dh_closed(x13):
    """3 x 13 Jacobian of ``h_closed`` in the column order
    [c(3) q(4) e(3) l(3)] (extended_kalman_filter.py:327-343, 347)."""
    x13 = np.asarray(x13, dtype=np.float64)
    c, q, lm = x13[0:3], x13[3:7], x13[10:13]
    a, u = q[0], q[1:4]
    v = lm - c
    s = a * a + u @ u
    uv = u @ v
    f = (a * a - u @ u) * v + 2.0 * uv * u - 2.0 * a * np.cross(u, v)
    h = f / s
    # R(q)^T = d h / d l
    rt = ((a * a - u @ u) * np.eye(3) + 2.0 * np.outer(u, u) - 2.0 * a * _skew(u)) / s
    g = np.empty((3, 4))
    g[:, 0] = 2.0 * a * v - 2.0 * np.cross(u, v)
    g[:, 1:4] = (-2.0 * np.outer(v, u) + 2.0 * np.outer(u, v)
                 + 2.0 * uv * np.eye(3) + 2.0 * a * _skew(v))
    dq = g / s - 2.0 * np.outer(h, q) / s
    e_map = np.empty((4, 3))
    e_map[0, :] = -u
    e_map[1:4, :] = a * np.eye(3) - _skew(u)
    out = np.empty((3, 13))
    out[:, 0:3] = -rt
    out[:, 3:7] = dq
    out[:, 7:10] = dq @ e_map
    out[:, 10:13] = rt
    return out


def rotmat_scalar_first(q):
    """Rotation matrix of the (normalised) scalar-first quaternion, as
    ``Rotation.from_quat(q, scalar_first=True).as_matrix()``
    (extended_kalman_filter.py:264-267)."""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.sqrt(q @ q)
    a, u = q[0], q[1:4]
    return (a * a - u @ u) * np.eye(3) + 2.0 * np.outer(u, u) + 2.0 * a * _skew(u)


def quat_update_as_written(q_stored, err):
    """The reference's camera-quaternion injection *as written*
    (extended_kalman_filter.py:138-149): the scalar-first arrays
    ``q = [qw qx qy qz]`` and ``dq = [1, e/2]`` are handed to SciPy's
    scalar-LAST ``Rotation.from_quat``, composed ``dq * q``, and written back
    with ``as_quat(scalar_first=True)``."""
    q = np.asarray(q_stored, dtype=np.float64)
    d = np.array([1.0, err[0] / 2.0, err[1] / 2.0, err[2] / 2.0])
    q = q / np.sqrt(q @ q)
    d = d / np.sqrt(d @ d)
    x1, y1, z1, w1 = d          # SciPy reads them as x, y, z, w
    x2, y2, z2, w2 = q
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    y = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2
    z = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2
    out = np.array([w, x, y, z])
    return out / np.sqrt(out @ out)


def quat_update_scalar_first(q_stored, err):
    """What the code evidently meant (and what ekf_with_rotations.py:150-151
    does): Hamilton product (1, e/2) * q with consistent scalar-first order."""
    q = np.asarray(q_stored, dtype=np.float64)
    d = np.array([1.0, err[0] / 2.0, err[1] / 2.0, err[2] / 2.0])
    q = q / np.sqrt(q @ q)
    d = d / np.sqrt(d @ d)
    w1, x1, y1, z1 = d
    w2, x2, y2, z2 = q
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    y = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2
    z = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2
    out = np.array([w, x, y, z])
    return out / np.sqrt(out @ out)


class OracleEKF:
    """Restatement of reference ``EKF`` (extended_kalman_filter.py:37-357)
    without the vision front-end.

    ``store_dtype=np.float32`` emulates an fp32-stored covariance (P rounded to
    fp32 after every write, all arithmetic still fp64): used to size the
    tolerance of the HIP fp32-covariance path.
    """

    def __init__(self, initial_camera_pose, mode="reference_ops",
                 quat_mode="as_written", store_dtype=np.float64):
        if mode not in ("reference_ops", "fast"):
            raise ValueError(mode)
        if quat_mode not in ("as_written", "scalar_first"):
            raise ValueError(quat_mode)
        self.mode = mode
        self.quat_mode = quat_mode
        self.store_dtype = store_dtype
        self.state = np.array(initial_camera_pose)            # :46 (keeps int64)
        self.uncertainty = np.eye(CAM) * INITIAL_CAMERA_UNCERTAINTY   # :48
        self.num_landmarks = 0
        self.landmarks = {}

    # -- boundary getters (extended_kalman_filter.py:84-93, 355-357) ---------
    def get_poses(self):
        return self.state[:CAM], self.state[CAM:].reshape(-1, LM)

    def get_lm_uncertainties(self):
        return np.diagonal(np.asarray(self.uncertainty))[CAM:].reshape(-1, LM)

    def get_lm_estimates(self):
        return self.landmarks.items()

    # -- extended_kalman_filter.py:58-82 -------------------------------------
    def observe(self, ids, poses):
        for idx, pose in zip(ids, poses):
            if idx not in self.landmarks:
                self.add_marker(idx, pose)
        self.predict()
        self.update(ids, poses)

    # -- extended_kalman_filter.py:239-290 -----------------------------------
    def add_marker(self, idx, pose, uncertainity=None):
        self.landmarks[idx] = self.num_landmarks
        self.num_landmarks += 1
        cam = self.state[:CAM]
        rot_mc = rotmat_scalar_first(cam[3:7])
        if self.mode == "reference_ops":
            rot_cm = np.linalg.inv(rot_mc)                    # :269
        else:
            rot_cm = rot_mc.T
        t_ml = rot_cm @ np.asarray(pose, dtype=np.float64)[0:3] + cam[0:3]
        self.state = np.hstack((self.state, t_ml))            # :274
        n_dims = LM * self.num_landmarks + CAM
        grown = np.zeros((n_dims, n_dims))
        grown[: n_dims - LM, : n_dims - LM] = np.asarray(self.uncertainty)
        block = (np.full(LM, INITIAL_LANDMARK_UNCERTAINTY) if uncertainity is None
                 else np.asarray(uncertainity, dtype=np.float64) * np.ones(LM))
        grown[n_dims - LM:, n_dims - LM:] = np.diag(block)    # :279-285
        self.uncertainty = self._store(grown)

    def _store(self, p):
        if self.store_dtype == np.float64:
            return p
        return np.asarray(p).astype(self.store_dtype).astype(np.float64)

    def process_noise_diag(self):
        """Diagonal of Q (extended_kalman_filter.py:98-104)."""
        n_dims = LM * self.num_landmarks + CAM
        qd = np.full(n_dims, Q_UNCERTAINTY_LM)
        qd[0:3] = Q_UNCERTAINTY_CAM
        qd[3:7] = 0.0
        qd[7:10] = Q_ERROR_UNCERTAINTY_CAM
        return qd

    # -- extended_kalman_filter.py:95-105 ------------------------------------
    def predict(self):
        qd = self.process_noise_diag()
        if self.mode == "reference_ops":
            n_dims = qd.shape[0]
            qfull = np.zeros((n_dims, n_dims))                # dense, as :99
            qfull[np.arange(n_dims), np.arange(n_dims)] = qd
            self.uncertainty = self.uncertainty + qfull
        else:
            p = np.array(self.uncertainty, dtype=np.float64)
            p[np.arange(p.shape[0]), np.arange(p.shape[0])] += qd
            self.uncertainty = p
        # (no _store here: predict+update is one write of P in the HIP path)

    # -- extended_kalman_filter.py:158-237 -----------------------------------
    def measurement_blocks(self, ids, poses):
        """z (k,), h (k,), per-landmark 3x13 Jacobians (m,3,13) and the first
        state column of every visible landmark (m,)."""
        m = len(ids)
        z = np.empty(3 * m)
        hv = np.empty(3 * m)
        jac = np.empty((m, 3, 13))
        col = np.empty(m, dtype=np.int64)
        cam = np.asarray(self.state[:CAM], dtype=np.float64)
        for j, (idx, pose) in enumerate(zip(ids, poses)):
            index = self.landmarks[idx]
            c0 = LM * index + CAM
            x13 = np.concatenate((cam, self.state[c0:c0 + LM]))
            z[3 * j:3 * j + 3] = np.asarray(pose, dtype=np.float64)[0:3]   # :192,196
            hv[3 * j:3 * j + 3] = h_closed(x13)
            jac[j] = dh_closed(x13)
            col[j] = c0
        return z, hv, jac, col

    def dense_jacobian(self, jac, col):
        n_dims = LM * self.num_landmarks + CAM
        dh = np.zeros((3 * len(col), n_dims))
        for j, c0 in enumerate(col):
            dh[3 * j:3 * j + 3, 0:CAM] = jac[j][:, 0:CAM]      # :233
            dh[3 * j:3 * j + 3, c0:c0 + LM] = jac[j][:, CAM:]  # :236
        return dh

    # -- extended_kalman_filter.py:107-156 -----------------------------------
    def update(self, ids, poses):
        z, hv, jac, col = self.measurement_blocks(ids, poses)
        dh = self.dense_jacobian(jac, col)
        resid = z - hv
        k = dh.shape[0]
        if self.mode == "reference_ops":
            from scipy import sparse
            from scipy.sparse.linalg import spsolve
            dh_s = sparse.csr_matrix(dh)                       # :121
            p_s = sparse.csr_matrix(self.uncertainty)          # :122
            s = dh_s @ p_s @ dh_s.T + sparse.eye(k, format="csc") * R_UNCERTAINTY
            s_inv = spsolve(sparse.csc_matrix(s), sparse.eye(k, format="csc"))
            gain = p_s @ dh_s.T @ s_inv                        # :130
            delta = np.asarray(gain @ resid).ravel()           # :131
            n_dims = dh.shape[1]
            new_p = (np.eye(n_dims) - gain @ dh_s) @ self.uncertainty   # :155-156
            new_p = np.asarray(new_p)
        else:
            p = np.asarray(self.uncertainty)
            hp = dh @ p                                        # k x N
            pht = p @ dh.T                                     # N x k
            s = hp @ dh.T + R_UNCERTAINTY * np.eye(k)
            chol = np.linalg.cholesky(0.5 * (s + s.T))
            w_right = np.linalg.solve(chol, hp)                # L^-1 (H P)
            w_left = np.linalg.solve(chol, pht.T)              # L^-1 (P H^T)^T
            y = np.linalg.solve(chol, resid)
            delta = w_left.T @ y
            new_p = p - w_left.T @ w_right
        self.intermediates = {"z": z, "h": hv, "jac": jac, "col": col,
                              "S": np.asarray(s.todense()) if hasattr(s, "todense") else s,
                              "delta": delta}
        self._inject(delta)
        self.uncertainty = self._store(new_p)

    def _inject(self, delta):
        """State injection, extended_kalman_filter.py:133-152 (delta[3:7] is
        dropped, every landmark moves, error state is reset)."""
        st = np.asarray(self.state, dtype=np.float64).copy()
        st[0:3] += delta[0:3]
        st[CAM:] += delta[CAM:]
        if self.quat_mode == "as_written":
            st[3:7] = quat_update_as_written(st[3:7], delta[7:10])
        else:
            st[3:7] = quat_update_scalar_first(st[3:7], delta[7:10])
        st[7:10] = 0.0
        self.state = st


# ==========================================================================
# EKF_Rotations  (/root/reference/filters/ekf_with_rotations.py:43-431)
# landmark = [x y z | qw qx qy qz | ex ey ez] (10 dims), 7 measurement rows per
# detection [xyz_cl ; q_cl], Jacobian 7 x 20 in the order
# [c(3) qc(4) ec(3) | l(3) ql(4) el(3)]  (:392-415).
# ==========================================================================
ROT_Q_UNCERTAINTY_CAM = 0.2          # ekf_with_rotations.py:27
ROT_LM = 10
ROT_ROWS = 7


def _qmul(a, b):
    """Hamilton product, scalar-first."""
    aw, av = a[0], a[1:4]
    bw, bv = b[0], b[1:4]
    return np.concatenate(([aw * bw - av @ bv], aw * bv + bw * av + np.cross(av, bv)))


def _lmat(a):
    """a (x) b = L(a) b."""
    w, x, y, z = a
    return np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])


def _rmat(b):
    """a (x) b = R(b) a."""
    w, x, y, z = b
    return np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])


def _emap(q):
    """d((1, e) (x) q)/de at e = 0  (4 x 3)."""
    a, u = q[0], q[1:4]
    out = np.empty((4, 3))
    out[0, :] = -u
    out[1:4, :] = a * np.eye(3) - _skew(u)
    return out


def h_rot_closed(x20):
    """[xyz_cl ; q_cl] with q_cl = (dq_c*q_c)^-1 (x) (dq_l*q_l) at e = 0
    (ekf_with_rotations.py:363-390; sympy's Quaternion.inverse() = conj / |q|^2)."""
    x20 = np.asarray(x20, dtype=np.float64)
    p, r = x20[3:7], x20[13:17]
    s = p @ p
    conj = p * np.array([1.0, -1.0, -1.0, -1.0])
    xyz = h_closed(np.concatenate((x20[0:10], x20[10:13])))
    return np.concatenate((xyz, _qmul(conj, r) / s))


def dh_rot_closed(x20):
    """7 x 20 Jacobian of ``h_rot_closed`` (ekf_with_rotations.py:420)."""
    x20 = np.asarray(x20, dtype=np.float64)
    p, r = x20[3:7], x20[13:17]
    s = p @ p
    conj = p * np.array([1.0, -1.0, -1.0, -1.0])
    qcl = _qmul(conj, r) / s
    out = np.zeros((7, 20))
    j13 = dh_closed(np.concatenate((x20[0:10], x20[10:13])))
    out[0:3, 0:10] = j13[:, 0:10]
    out[0:3, 10:13] = j13[:, 10:13]
    dp = _rmat(r) @ np.diag([1.0, -1.0, -1.0, -1.0]) / s - 2.0 * np.outer(qcl, p) / s
    dr = _lmat(conj) / s
    out[3:7, 3:7] = dp
    out[3:7, 7:10] = dp @ _emap(p)
    out[3:7, 13:17] = dr
    out[3:7, 17:20] = dr @ _emap(r)
    return out


def quat_from_euler_xyz(angles):
    """``Rotation.from_euler("xyz", a).as_quat(scalar_first=True)``: extrinsic rotations about
    x, then y, then z, i.e. q = qz (x) qy (x) qx  (ekf_with_rotations.py:216-219)."""
    a, b, c = 0.5 * np.asarray(angles, dtype=np.float64)
    qx = np.array([np.cos(a), np.sin(a), 0.0, 0.0])
    qy = np.array([np.cos(b), 0.0, np.sin(b), 0.0])
    qz = np.array([np.cos(c), 0.0, 0.0, np.sin(c)])
    return _qmul(qz, _qmul(qy, qx))


def quat_from_matrix(mat):
    """SciPy's ``Rotation.from_matrix(M).as_quat(scalar_first=True)`` (no sign
    canonicalisation): branch on the largest of (M00, M11, M22, trace)."""
    m = np.asarray(mat, dtype=np.float64)
    dec = np.array([m[0, 0], m[1, 1], m[2, 2], m[0, 0] + m[1, 1] + m[2, 2]])
    ch = int(np.argmax(dec))
    q = np.empty(4)                                   # x y z w
    if ch != 3:
        i, j, k = ch, (ch + 1) % 3, (ch + 2) % 3
        q[i] = 1.0 - dec[3] + 2.0 * m[i, i]
        q[j] = m[j, i] + m[i, j]
        q[k] = m[k, i] + m[i, k]
        q[3] = m[k, j] - m[j, k]
    else:
        q[0] = m[2, 1] - m[1, 2]
        q[1] = m[0, 2] - m[2, 0]
        q[2] = m[1, 0] - m[0, 1]
        q[3] = 1.0 + dec[3]
    q = q / np.sqrt(q @ q)
    return np.array([q[3], q[0], q[1], q[2]])


def rotmat_euler_xyz(angles):
    a, b, c = np.asarray(angles, dtype=np.float64)
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    rz = np.array([[cc, -sc, 0], [sc, cc, 0], [0, 0, 1]])
    return rz @ ry @ rx


class OracleEKFRotations:
    """Restatement of reference ``EKF_Rotations`` (ekf_with_rotations.py:43-431)."""

    def __init__(self, initial_camera_pose, mode="reference_ops", store_dtype=np.float64):
        if mode not in ("reference_ops", "fast"):
            raise ValueError(mode)
        self.mode = mode
        self.store_dtype = store_dtype
        self.state = np.array(initial_camera_pose)
        self.uncertainty = np.eye(CAM) * INITIAL_CAMERA_UNCERTAINTY      # :54
        self.num_landmarks = 0
        self.landmarks = {}

    def get_poses(self):
        return self.state[:CAM], self.state[CAM:].reshape(-1, ROT_LM)

    def get_lm_uncertainties(self):
        return np.diagonal(np.asarray(self.uncertainty))[CAM:].reshape(-1, ROT_LM)

    def get_lm_estimates(self):
        return self.landmarks.items()

    def _store(self, p):
        if self.store_dtype == np.float64:
            return p
        return np.asarray(p).astype(self.store_dtype).astype(np.float64)

    # -- :66-90 ----------------------------------------------------------------
    def observe(self, ids, poses):
        for idx, pose in zip(ids, poses):
            if idx not in self.landmarks:
                self.add_marker(idx, pose)
        self.predict()
        self.update(ids, poses)

    # -- :275-335 --------------------------------------------------------------
    def add_marker(self, idx, pose, uncertainity=None):
        pose = np.asarray(pose, dtype=np.float64)
        self.landmarks[idx] = self.num_landmarks
        self.num_landmarks += 1
        cam = self.state[:CAM]
        rot_mc = rotmat_scalar_first(cam[3:7])
        rot_cm = np.linalg.inv(rot_mc) if self.mode == "reference_ops" else rot_mc.T
        rot_cl = rotmat_euler_xyz(pose[3:6])          # pose[QUAT_DIMS] on a 6-vector = rvec (:307-310)
        t_ml = rot_cm @ pose[0:3] + cam[0:3]
        q_ml = quat_from_matrix(rot_cm @ rot_cl)
        self.state = np.hstack((self.state, t_ml, q_ml, np.zeros(3)))
        n_dims = ROT_LM * self.num_landmarks + CAM
        grown = np.zeros((n_dims, n_dims))
        grown[: n_dims - ROT_LM, : n_dims - ROT_LM] = np.asarray(self.uncertainty)
        block = (np.full(ROT_LM, INITIAL_LANDMARK_UNCERTAINTY) if uncertainity is None
                 else np.asarray(uncertainity, dtype=np.float64) * np.ones(ROT_LM))
        grown[n_dims - ROT_LM:, n_dims - ROT_LM:] = np.diag(block)
        self.uncertainty = self._store(grown)

    def process_noise_diag(self):
        """:103-113 -- every landmark dimension gets 0.01 (Q_UNCERTAINTY_LM_QUAT is unused)."""
        n_dims = ROT_LM * self.num_landmarks + CAM
        qd = np.full(n_dims, Q_UNCERTAINTY_LM)
        qd[0:3] = ROT_Q_UNCERTAINTY_CAM
        qd[3:7] = 0.0
        qd[7:10] = Q_ERROR_UNCERTAINTY_CAM
        return qd

    def predict(self):
        qd = self.process_noise_diag()
        p = np.array(self.uncertainty, dtype=np.float64)
        p[np.arange(p.shape[0]), np.arange(p.shape[0])] += qd
        self.uncertainty = p

    # -- :183-273 --------------------------------------------------------------
    def measurement_blocks(self, ids, poses):
        m = len(ids)
        z = np.empty(ROT_ROWS * m)
        hv = np.empty(ROT_ROWS * m)
        jac = np.empty((m, ROT_ROWS, 20))
        col = np.empty(m, dtype=np.int64)
        cam = np.asarray(self.state[:CAM], dtype=np.float64)
        for j, (idx, pose) in enumerate(zip(ids, poses)):
            pose = np.asarray(pose, dtype=np.float64)
            c0 = ROT_LM * self.landmarks[idx] + CAM
            x20 = np.concatenate((cam, self.state[c0:c0 + ROT_LM]))
            z[7 * j:7 * j + 3] = pose[0:3]
            z[7 * j + 3:7 * j + 7] = quat_from_euler_xyz(pose[3:6])
            hv[7 * j:7 * j + 7] = h_rot_closed(x20)
            jac[j] = dh_rot_closed(x20)
            col[j] = c0
        return z, hv, jac, col

    def dense_jacobian(self, jac, col):
        n_dims = ROT_LM * self.num_landmarks + CAM
        dh = np.zeros((ROT_ROWS * len(col), n_dims))
        for j, c0 in enumerate(col):
            dh[7 * j:7 * j + 7, 0:CAM] = jac[j][:, 0:CAM]
            dh[7 * j:7 * j + 7, c0:c0 + ROT_LM] = jac[j][:, CAM:]
        return dh

    # -- :115-181 --------------------------------------------------------------
    def update(self, ids, poses):
        z, hv, jac, col = self.measurement_blocks(ids, poses)
        dh = self.dense_jacobian(jac, col)
        resid = z - hv
        k = dh.shape[0]
        p = np.asarray(self.uncertainty)
        if self.mode == "reference_ops":
            from scipy import sparse
            from scipy.sparse.linalg import spsolve
            dh_s = sparse.csr_matrix(dh)
            p_s = sparse.csr_matrix(p)
            s = dh_s @ p_s @ dh_s.T + sparse.eye(k, format="csc") * R_UNCERTAINTY
            s_inv = spsolve(sparse.csc_matrix(s), sparse.eye(k, format="csc"))
            gain = p_s @ dh_s.T @ s_inv
            delta = np.asarray(gain @ resid).ravel()
            new_p = np.asarray((np.eye(dh.shape[1]) - gain @ dh_s) @ p)
        else:
            hp = dh @ p
            pht = p @ dh.T
            s = hp @ dh.T + R_UNCERTAINTY * np.eye(k)
            chol = np.linalg.cholesky(0.5 * (s + s.T))
            delta = np.linalg.solve(chol, pht.T).T @ np.linalg.solve(chol, resid)
            new_p = p - np.linalg.solve(chol, pht.T).T @ np.linalg.solve(chol, hp)
        self._inject(delta)
        self.uncertainty = self._store(new_p)

    def _inject(self, delta):
        """:142-177: xyz additive, quaternions multiplicative with the consistent scalar-first
        convention, for the camera and for EVERY landmark; the landmarks' error states stay 0."""
        st = np.asarray(self.state, dtype=np.float64).copy()
        st[0:3] += delta[0:3]
        st[3:7] = quat_update_scalar_first(st[3:7], delta[7:10])
        st[7:10] = 0.0
        for i in range(self.num_landmarks):
            c0 = ROT_LM * i + CAM
            st[c0:c0 + 3] += delta[c0:c0 + 3]
            st[c0 + 3:c0 + 7] = quat_update_scalar_first(st[c0 + 3:c0 + 7], delta[c0 + 7:c0 + 10])
        self.state = st
