"""``EKF``: drop-in for the reference filter of the same name
(/root/reference/filters/extended_kalman_filter.py:37-357) whose per-frame
``observe`` (predict + update, :58-156) runs as HIP kernels on an MI355X.

Same constructor, methods, attributes and error behaviour at the boundary
``BaseFilter`` drives.  The marker-id -> landmark-index
dictionary stays here on the host; state and covariance live in HBM.
"""
from __future__ import annotations

import numpy as np

from ..hip_backend import HipEkf
from .base_filter import BaseFilter, dictionary_size

MOVING_AVG_WINDOW = 10   # unused in the reference as well (:19)

INITIAL_CAMERA_UNCERTAINTY = 0.1
INITIAL_LANDMARK_UNCERTAINTY = 0.7
R_UNCERTAINTY = 0.9
Q_UNCERTAINTY_CAM = 0.3
Q_ERROR_UNCERTAINTY_CAM = 0.5
Q_UNCERTAINTY_LM = 0.01

CAM_DIMS = 10
XYZ_DIMS = slice(0, 3)
QUAT_DIMS = slice(3, 7)
ERROR_DIMS = slice(7, 10)
LM_DIMS = 3


class EKF(BaseFilter):
    """Object for tracking the positions of the camera and landmarks."""

    def __init__(self, initial_camera_pose, aruco_dict=None, *, max_landmarks: int | None = None,
                 max_visible: int | None = None, cov_dtype: str = "float64",
                 quat_update: str = "as_written", cov_kernel: str = "auto",
                 device: str = "cuda:0", map_file=None, lookahead: bool | None = None,
                 fused: bool = True) -> None:
        """Positional arguments as the reference (:40-43).  Keyword-only extras:
        initial capacity (default: the number of ids of ``aruco_dict`` -- DICT_5X5_50
        has 50, base_filter.py:81-82; the buffers grow when more markers or more
        detections per frame show up, as the reference's arrays do, :274-290), covariance
        storage dtype, and the quaternion-injection convention
        (``"as_written"`` reproduces :138-149 exactly, ``"scalar_first"`` is the
        consistent one)."""
        super().__init__(initial_camera_pose, map_file, aruco_dict)
        self._initial_pose = np.array(initial_camera_pose)      # :46, dtype kept
        if self._initial_pose.shape != (CAM_DIMS,):
            raise ValueError("initial_camera_pose must have 10 entries")
        self.num_landmarks = 0
        self.landmarks = {}
        if max_landmarks is None:
            max_landmarks = dictionary_size(aruco_dict)
        if max_visible is None:
            max_visible = min(max_landmarks, 64)
        self._hip = HipEkf(max_landmarks, max_visible, cov_dtype=cov_dtype,
                           quat_mode=quat_update, cov_kernel=cov_kernel, device=device, lookahead=lookahead, fused=fused,
                           noise={"initial_camera_uncertainty": INITIAL_CAMERA_UNCERTAINTY,
                                  "initial_landmark_uncertainty": INITIAL_LANDMARK_UNCERTAINTY,
                                  "r_uncertainty": R_UNCERTAINTY, "q_cam": Q_UNCERTAINTY_CAM,
                                  "q_err": Q_ERROR_UNCERTAINTY_CAM, "q_lm": Q_UNCERTAINTY_LM})
        self._hip.reset(self._initial_pose.astype(np.float64))
        self._load_initial_map()

    # -- attributes the base class / callers read (base_filter.py:293-304) ---
    @property
    def state(self) -> np.ndarray:
        if self.num_landmarks == 0:
            # the reference keeps the caller's (int64) array until the first
            # hstack (:46, :274): D7 in SURVEY appendix A
            return self._initial_pose
        return self._hip.get_state()

    @property
    def uncertainty(self) -> np.ndarray:
        return self._hip.get_cov()

    # -- :58-82 ----------------------------------------------------------------
    def observe(self, ids, poses) -> None:
        """Add unseen markers, predict, update.  ``ids``: iterable of marker
        ids; ``poses``: (m, 6) ``[tvec | rvec]``, only ``pose[0:3]`` is used."""
        if isinstance(ids, np.ndarray) and ids.dtype.kind in "iu":
            ids = ids.reshape(-1).tolist()     # (what process_frame passes: ids.flatten(), base_filter.py:199)
        else:
            ids = [int(i) for i in ids]
        if not ids:
            raise ValueError("observe() needs at least one detection")
        poses = np.asarray(poses, dtype=np.float64).reshape(len(ids), -1)
        known = self.landmarks
        try:                                   # steady state: every marker of the frame is in the map already
            index = [known[i] for i in ids]
        except KeyError:
            fresh, new_xyz = [], []
            for idx, pose in zip(ids, poses):
                if idx in known or idx in fresh:
                    continue
                fresh.append(idx)
                new_xyz.append(pose[XYZ_DIMS])
            # every add_marker of a frame sees the same (pre-update) camera pose
            self._hip.add_markers(np.asarray(new_xyz))
            for idx in fresh:
                known[idx] = self.num_landmarks
                self.num_landmarks += 1
            index = [known[i] for i in ids]
        self._hip.observe(index, poses[:, XYZ_DIMS])

    # -- :239-290 --------------------------------------------------------------
    def add_marker(self, idx, pose, uncertainity=None) -> None:
        """(sic) ``uncertainity`` -- keyword spelled as in the reference."""
        self._hip.add_markers(np.asarray(pose, dtype=np.float64)[XYZ_DIMS][None, :], uncertainity)
        self.landmarks[idx] = self.num_landmarks
        self.num_landmarks += 1

    # -- :84-93, :355-357 --------------------------------------------------------
    def get_poses(self):
        state = self.state
        return state[:CAM_DIMS], state[CAM_DIMS:].reshape(-1, LM_DIMS)

    def get_lm_uncertainties(self) -> np.ndarray:
        return self._hip.get_cov_diag()[CAM_DIMS:].reshape(-1, LM_DIMS)

    def get_lm_estimates(self):
        return self.landmarks.items()

    # -- extras: resident-detection batch entry, used by bench / replay --------
    @property
    def backend(self) -> HipEkf:
        return self._hip
