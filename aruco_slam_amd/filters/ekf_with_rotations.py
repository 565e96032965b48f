"""``EKF_Rotations``: drop-in for the reference filter of the same name
(/root/reference/filters/ekf_with_rotations.py:43-431) on the MI355X kernels.

Same pipeline as ``EKF`` with the landmark orientation in the state: landmark =
``[x y z | qw qx qy qz | ex ey ez]`` (10 dims, state ``10 n + 10``), 7 measurement rows per
detection ``[xyz_cl ; q_cl]`` with ``q_cl = (dq_c q_c)^-1 (dq_l q_l)`` (:363-390), additive
residual also on the quaternion components (:140), multiplicative scalar-first quaternion
updates for the camera and every landmark (:146-177), process noise cam 0.2 / err 0.5 / every
landmark dimension 0.01 (:26-31, 103-113).  Quirks kept: ``rvec`` is read as extrinsic "xyz"
Euler angles (:216-219, 307-310).
"""
from __future__ import annotations

import numpy as np

from ..hip_backend import HipEkf
from .base_filter import BaseFilter, dictionary_size

INITIAL_CAMERA_UNCERTAINTY = 0.1
INITIAL_LANDMARK_UNCERTAINTY = 0.7
R_UNCERTAINTY = 0.9
Q_UNCERTAINTY_CAM = 0.2
Q_ERROR_UNCERTAINTY_CAM = 0.5
Q_UNCERTAINTY_LM_XYZ = 0.01
Q_UNCERTAINTY_LM_QUAT = 0.05     # unused in the reference as well (:31)

CAM_DIMS = 10
XYZ_DIMS = slice(0, 3)
LM_DIMS = 10


def euler_xyz_to_quat(angles) -> np.ndarray:
    """``Rotation.from_euler("xyz", a).as_quat(scalar_first=True)`` for an ``(m, 3)`` array:
    extrinsic x, y, z rotations, q = qz * qy * qx (Hamilton), scalar first."""
    a = 0.5 * np.asarray(angles, dtype=np.float64).reshape(-1, 3)
    cx, sx, cy, sy, cz, sz = np.cos(a[:, 0]), np.sin(a[:, 0]), np.cos(a[:, 1]), np.sin(a[:, 1]), \
        np.cos(a[:, 2]), np.sin(a[:, 2])
    # qy * qx = (cy cx, cy sx, sy cx, -sy sx);  then qz * (.)
    w1, x1, y1, z1 = cy * cx, cy * sx, sy * cx, -sy * sx
    return np.stack([cz * w1 - sz * z1, cz * x1 - sz * y1, cz * y1 + sz * x1, cz * z1 + sz * w1], axis=1)


class EKF_Rotations(BaseFilter):  # noqa: N801  (name of the reference class)
    """Object for tracking the poses of the camera and of the landmarks."""

    def __init__(self, initial_camera_pose, *, max_landmarks: int | None = None, max_visible: int | None = None,
                 cov_dtype: str = "float64", cov_kernel: str = "auto", device: str = "cuda:0",
                 lookahead: bool | None = None, fused: bool = True) -> None:
        super().__init__(initial_camera_pose, None)
        self._initial_pose = np.array(initial_camera_pose)
        if self._initial_pose.shape != (CAM_DIMS,):
            raise ValueError("initial_camera_pose must have 10 entries")
        self.num_landmarks = 0
        self.landmarks = {}
        if max_landmarks is None:
            max_landmarks = dictionary_size(None)    # (the reference's EKF_Rotations takes no aruco_dict: DICT_5X5_50)
        if max_visible is None:
            max_visible = min(max_landmarks, 27)     # 7 rows per detection, k <= 192: the fused front kernel (grows to 50)
        self._hip = HipEkf(max_landmarks, max_visible, cov_dtype=cov_dtype, quat_mode="scalar_first",
                           cov_kernel=cov_kernel, device=device, lookahead=lookahead, fused=fused,
                           model="ekf_rotations",
                           noise={"initial_camera_uncertainty": INITIAL_CAMERA_UNCERTAINTY,
                                  "initial_landmark_uncertainty": INITIAL_LANDMARK_UNCERTAINTY,
                                  "r_uncertainty": R_UNCERTAINTY, "q_cam": Q_UNCERTAINTY_CAM,
                                  "q_err": Q_ERROR_UNCERTAINTY_CAM, "q_lm": Q_UNCERTAINTY_LM_XYZ})
        self._hip.reset(self._initial_pose.astype(np.float64))

    @property
    def state(self) -> np.ndarray:
        if self.num_landmarks == 0:
            return self._initial_pose
        return self._hip.get_state()

    @property
    def uncertainty(self) -> np.ndarray:
        return self._hip.get_cov()

    # -- :66-90 ----------------------------------------------------------------
    def observe(self, ids, poses) -> None:
        if isinstance(ids, np.ndarray) and ids.dtype.kind in "iu":
            ids = ids.reshape(-1).tolist()
        else:
            ids = [int(i) for i in ids]
        if not ids:
            raise ValueError("observe() needs at least one detection")
        poses = np.asarray(poses, dtype=np.float64).reshape(len(ids), -1)
        known = self.landmarks
        try:                                   # steady state: every marker of the frame is in the map already
            index = [known[i] for i in ids]
        except KeyError:
            fresh, new_pose = [], []
            for idx, pose in zip(ids, poses):
                if idx in known or idx in fresh:
                    continue
                fresh.append(idx)
                new_pose.append(pose[:6])
            self._hip.add_markers(np.asarray(new_pose))
            for idx in fresh:
                known[idx] = self.num_landmarks
                self.num_landmarks += 1
            index = [known[i] for i in ids]
        z = np.hstack((poses[:, XYZ_DIMS], euler_xyz_to_quat(poses[:, 3:6])))      # :216-224
        self._hip.observe(index, z)

    # -- :275-335 --------------------------------------------------------------
    def add_marker(self, idx, pose, uncertainity=None) -> None:
        self._hip.add_markers(np.asarray(pose, dtype=np.float64)[:6][None, :], uncertainity)
        self.landmarks[idx] = self.num_landmarks
        self.num_landmarks += 1

    # -- :92-101, :429-431 -------------------------------------------------------
    def get_poses(self):
        state = self.state
        return state[:CAM_DIMS], state[CAM_DIMS:].reshape(-1, LM_DIMS)

    def get_lm_uncertainties(self) -> np.ndarray:
        return self._hip.get_cov_diag()[CAM_DIMS:].reshape(-1, LM_DIMS)

    def get_lm_estimates(self):
        return self.landmarks.items()

    @property
    def backend(self) -> HipEkf:
        return self._hip
