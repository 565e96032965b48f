"""Filter boundary: host-side mirror of reference ``BaseFilter``
(/root/reference/filters/base_filter.py:35-381).

Only the boundary is in scope: the abstract filter API,
the ``process_frame`` driver contract (observe only when something was
detected, ``get_poses`` every frame -- base_filter.py:194-212) and the map file
format (``save_map`` :214-247).  The ArUco detector is OpenCV work and is not
re-implemented (used as the reference does when ``cv2`` is importable; otherwise
frames of pre-computed detections are fed through ``process_detections``); the
per-marker solvePnP that follows it is one batched HIP kernel
(``estimate_pose_of_markers``).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

try:  # the image has no OpenCV; the boundary does not need it
    import cv2  # type: ignore
except ImportError:  # pragma: no cover
    cv2 = None

CALIB_MTX_FILE = "./calibration/camera_matrix.npy"      # base_filter.py:12
DIST_COEFFS_FILE = "./calibration/dist_coeffs.npy"      # base_filter.py:13

KALMAN_FILTER = "ekf"
FACTOR_GRAPH = "factorgraph"

NOT_IMPLEMENTED_ERROR = """
                        This method is not implemented in the base class and
                        should be implemented in a subclass.
                        """

CAM_DIMS = 10
XYZ_DIMS = slice(0, 3)
QUAT_DIMS = slice(3, 7)
ERROR_DIMS = slice(7, 10)

# markers in the predefined dictionaries, by OpenCV's enumeration (cv2.aruco.DICT_4X4_50 = 0 ... DICT_ARUCO_MIP_36h12 = 21):
# how many landmarks a map built with that dictionary can hold (the filters size their first buffers for it; they grow
# if more show up).  The reference's default is DICT_5X5_50 (base_filter.py:81-82).
ARUCO_DICT_SIZES = {0: 50, 1: 100, 2: 250, 3: 1000, 4: 50, 5: 100, 6: 250, 7: 1000, 8: 50, 9: 100, 10: 250, 11: 1000,
                    12: 50, 13: 100, 14: 250, 15: 1000, 16: 1024, 17: 30, 18: 35, 19: 2320, 20: 587, 21: 250}
DEFAULT_ARUCO_DICT = 4      # cv2.aruco.DICT_5X5_50


def dictionary_size(aruco_dict) -> int:
    """Number of marker ids of the dictionary the filter was constructed with (None: the reference's default)."""
    if aruco_dict is None:
        aruco_dict = DEFAULT_ARUCO_DICT
    try:
        return ARUCO_DICT_SIZES[int(aruco_dict)]
    except (KeyError, TypeError, ValueError):
        return ARUCO_DICT_SIZES[DEFAULT_ARUCO_DICT]


class BaseFilter:
    """Front-end + abstract back-end API (names and semantics of the reference)."""

    def __init__(self, initial_pose, map_file=None, aruco_dict=None) -> None:
        self.calib_matrix = None
        self.dist_coeffs = None
        self.detector = None
        if cv2 is not None:
            # same contract as base_filter.py:55-67
            if not Path(CALIB_MTX_FILE).exists():
                raise FileNotFoundError("Camera matrix not found. Run calibration.py first.")
            if not Path(DIST_COEFFS_FILE).exists():
                raise FileNotFoundError(
                    "Distortion coefficients not found. Run calibration.py first.")
            self.calib_matrix = np.load(CALIB_MTX_FILE)
            self.dist_coeffs = np.load(DIST_COEFFS_FILE)
            self.detector = self.init_aruco_detector(aruco_dict)
        self.camera_pose = initial_pose
        self._map_file = map_file

    def _load_initial_map(self):
        """Subclasses call this once their back-end exists (the reference calls
        load_map from the base ctor, base_filter.py:71-72, before the subclass
        state exists -- one reason its load_map never worked)."""
        if self._map_file is not None:
            self.load_map(self._map_file)

    def init_aruco_detector(self, aruco_dict):
        """base_filter.py:74-90 (needs OpenCV)."""
        if cv2 is None:
            raise RuntimeError("OpenCV (cv2) is not available: feed detections via "
                               "process_detections()")
        if aruco_dict is None:
            aruco_dict = cv2.aruco.DICT_5X5_50
        aruco_dict = cv2.aruco.getPredefinedDictionary(aruco_dict)
        params = cv2.aruco.DetectorParameters()
        params.cornerRefinementMethod = cv2.aruco.CORNER_REFINE_SUBPIX
        params.cornerRefinementWinSize = 3
        params.cornerRefinementMaxIterations = 3
        params.adaptiveThreshWinSizeMin = 3
        params.adaptiveThreshWinSizeMax = 30
        return cv2.aruco.ArucoDetector(aruco_dict, params)

    def estimate_pose_of_markers(self, corners, ids, marker_size):
        """base_filter.py:92-171: IPPE-square PnP of every detected marker -> (m,6) [tvec|rvec].  The reference
        loops over the markers with cv2.solvePnP(flags=SOLVEPNP_IPPE_SQUARE); here all markers of the frame go
        through one HIP kernel (csrc/ekf_pose_ippe.hip) with the same object-point order (:113-121), camera matrix
        and distortion coefficients.  `corners` as the detector returns them (one [1,4,2] array per marker)."""
        if self.calib_matrix is None:
            raise RuntimeError("no camera calibration: set calib_matrix / dist_coeffs (calibration/*.npy)")
        from aruco_slam_amd import hip_backend
        if len(ids) == 0:
            return np.zeros((0, 6))
        backend = getattr(self, "_hip", None)        # (the filter's own GPU: one sequence per GPU)
        device = str(backend.device) if backend is not None else "cuda:0"
        return hip_backend.estimate_poses(corners, marker_size, self.calib_matrix, self.dist_coeffs, device=device)

    def process_frame(self, frame, should_filter=True, iteration=0, marker_size=0.16):
        """base_filter.py:173-212."""
        if self.detector is None:
            raise RuntimeError("no ArUco detector (cv2 missing): use process_detections()")
        corners, ids, _ = tuple(self.detector.detectMarkers(frame))
        detected_poses = np.array([])
        if ids is not None:
            frame = cv2.aruco.drawDetectedMarkers(frame, corners, ids)
            ids = ids.flatten()
            detected_poses = self.estimate_pose_of_markers(corners, ids, marker_size)
        _, camera_pose, marker_poses, detected_poses = self.process_detections(
            ids, detected_poses, should_filter, iteration)
        return frame, camera_pose, marker_poses, detected_poses

    def process_detections(self, ids, detected_poses, should_filter=True, iteration=0):
        """The part of ``process_frame`` behind the detector
        (base_filter.py:196-212): ``ids`` is None for a frame without
        detections, in which case the filter is NOT stepped (no predict)."""
        if ids is None:
            detected_poses = np.array([])
        elif should_filter:
            self.observe(ids, detected_poses)
        if should_filter:
            camera_pose, marker_poses = self.get_poses()
        else:
            _, marker_poses = self.get_poses()
            camera_pose = self.get_cam_estimate(iteration)
        return None, camera_pose, marker_poses, detected_poses

    def save_map(self, filename: str) -> None:
        """Map text format of base_filter.py:214-247: three comment lines and a
        blank, then one record per landmark in index order -- marker id, pose
        numbers, their variances (each ``", "``-separated through ``str()``) and a
        blank line."""
        _, rows = self.get_poses()
        variances = self.get_lm_uncertainties()
        marker_of = dict((index, marker) for marker, index in self.get_lm_estimates())

        def record(index):
            pose = rows[index]
            numbers = (pose, variances[index, :len(pose)])
            return "\n".join([str(marker_of[index])] + [", ".join(str(v) for v in vals) for vals in numbers]) + "\n\n"

        header = "# landmark_id\n# x y z\n# uncertainty\n\n"
        Path(filename).write_text(header + "".join(record(i) for i in range(len(rows))), encoding="utf-8")

    def load_map(self, filename: str) -> None:
        """Reader for the ``save_map`` format (base_filter.py:249-272: skip 4
        header lines, stride 4).  The reference's version ends in
        ``self.filter.add_marker`` (an AttributeError, :272); this one calls the
        intended restore hook ``self.add_marker(id, pose, uncertainty)``."""
        with Path(filename).open("r", encoding="utf-8") as file:
            lines = file.readlines()[4:]
        for i in range(0, len(lines) - 2, 4):
            id_ = int(lines[i].strip())
            pose = np.array(lines[i + 1].strip().split(", "), np.float64)
            uncertainty = np.array(lines[i + 2].strip().split(", "), np.float64)
            self.add_marker(id_, pose, uncertainty)

    def reset(self) -> None:
        """Back to the state of a freshly constructed filter (initial pose, no landmarks, status cleared): the way out
        of a sticky device-side error (non-SPD innovation covariance, bad device-resident index, ...), usually followed
        by ``load_checkpoint``."""
        self.backend.reset(np.asarray(self._initial_pose, dtype=np.float64))
        self.landmarks = {}
        self.num_landmarks = 0

    # -- resume (no reference counterpart: its map restore is the dead :249-272) -----
    def save_checkpoint(self, filename: str) -> None:
        """Full filter state for an exact resume: state vector, dense covariance (float64 copy of
        the device matrix) and the marker-id -> index table, as a plain ``.npz``."""
        ids = [k for k, _ in sorted(self.get_lm_estimates(), key=lambda kv: kv[1])]
        np.savez(filename, state=np.asarray(self.state, dtype=np.float64),
                 cov=np.asarray(self.uncertainty, dtype=np.float64),
                 marker_ids=np.asarray(ids, dtype=np.int64), filter=type(self).__name__)

    def load_checkpoint(self, filename: str) -> None:
        """Inverse of ``save_checkpoint`` on a filter of the same class; the device covariance
        is overwritten bit-for-bit when the stored values fit the covariance dtype."""
        with np.load(filename, allow_pickle=False) as ck:
            if str(ck["filter"]) != type(self).__name__:
                raise ValueError(f"checkpoint of {ck['filter']} loaded into {type(self).__name__}")
            state, cov, ids = ck["state"], ck["cov"], ck["marker_ids"]
        if len(ids) == 0:
            return
        self.backend.set_state_cov(state, cov)
        self.landmarks = {int(k): i for i, k in enumerate(ids)}
        self.num_landmarks = len(ids)

    # -- abstract back-end API, base_filter.py:327-381 ------------------------
    def observe(self, ids, poses) -> None:
        raise NotImplementedError(NOT_IMPLEMENTED_ERROR)

    def get_poses(self):
        raise NotImplementedError(NOT_IMPLEMENTED_ERROR)

    def get_lm_uncertainties(self):
        raise NotImplementedError(NOT_IMPLEMENTED_ERROR)

    def get_lm_estimates(self):
        raise NotImplementedError(NOT_IMPLEMENTED_ERROR)

    def get_cam_estimate(self, iteration: int):
        raise NotImplementedError(NOT_IMPLEMENTED_ERROR)
