"""App loop: counterpart of the reference's ``main/run_slam.py``
(/root/reference/main/run_slam.py:82-173) for the filters this package ships.

    python -m aruco_slam_amd.main.run_slam --video input_video.mp4 --filter ekf
    python -m aruco_slam_amd.main.run_slam --detections tests/golden/c1_detections.npz

Same CLI (``--video``, ``--filter``), same outputs (``outputs/trajectory.txt``
one line per frame, ``outputs/map.txt`` at exit).  The image has no OpenCV and
no video, so frames can also come from a detections replay file
(``--detections``): per frame a timestamp and the ``(ids, poses)`` the ArUco
front-end would have produced (``ids`` absent on frames without detections, in
which case the filter is not stepped -- base_filter.py:197-204).  The 2-D/3-D
viewers are GUI code and are not part of this package.
"""
from __future__ import annotations

import argparse
from pathlib import Path

import numpy as np

from ..filters.base_filter import BaseFilter, cv2
from ..filters.ekf_with_rotations import EKF_Rotations
from ..filters.extended_kalman_filter import EKF
from ..outputs.trajectory_writer import TrajectoryWriter

TRAJECTORY_TEXT_FILE = "outputs/trajectory.txt"     # run_slam.py:28
MAP_FILE = "outputs/map.txt"                        # run_slam.py:32
IMAGE_SIZE = 1920, 1080                             # run_slam.py:43


def init_tracker(filter_type: str, initial_pose: np.ndarray, **kwargs) -> BaseFilter:
    """run_slam.py:69-79.  The two EKF back-ends are accelerated; the factor graph is GTSAM."""
    if filter_type == "ekf":
        return EKF(initial_pose, **kwargs)
    if filter_type == "ekf_rotations":
        return EKF_Rotations(initial_pose, **kwargs)
    if filter_type == "factorgraph":
        raise NotImplementedError(
            f"filter '{filter_type}' is outside this package's scope (SURVEY section 8)")
    raise ValueError(f"Unknown filter type: {filter_type}")


def detection_frames(path: str):
    """Yield ``(timestamp_ms, ids | None, poses)`` from a replay ``.npz`` with
    arrays ids [sum m], poses [sum m, 6], offsets [F+1], timestamps_ms [F],
    has_detections [F]."""
    det = np.load(path, allow_pickle=False)
    offs = det["offsets"]
    for f in range(len(det["timestamps_ms"])):
        sl = slice(int(offs[f]), int(offs[f + 1]))
        ids = det["ids"][sl] if det["has_detections"][f] else None
        yield float(det["timestamps_ms"][f]), ids, det["poses"][sl]


def main(cmdline_args: argparse.Namespace) -> None:
    initial_pose = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])     # run_slam.py:85-88 (int64)
    tracker = init_tracker(cmdline_args.filter, initial_pose,
                           **getattr(cmdline_args, "filter_kwargs", {}))
    out_dir = Path(getattr(cmdline_args, "output_dir", "outputs"))
    out_dir.mkdir(parents=True, exist_ok=True)

    with TrajectoryWriter(str(out_dir / "trajectory.txt")) as cam_traj_writer:
        if cmdline_args.detections:
            for timestamp, ids, poses in detection_frames(cmdline_args.detections):
                _, camera_pose, _, _ = tracker.process_detections(ids, poses)
                cam_traj_writer.write(timestamp, camera_pose)
        else:
            if cv2 is None:
                raise RuntimeError("OpenCV (cv2) is not installed: pass --detections <replay.npz> "
                                   "instead of --video")
            cap = cv2.VideoCapture(cmdline_args.video)
            cap.set(cv2.CAP_PROP_BUFFERSIZE, 0)
            for _ in range(int(cap.get(cv2.CAP_PROP_FRAME_COUNT))):
                ret, frame = cap.read()
                if not ret:
                    break
                frame = cv2.resize(frame, IMAGE_SIZE)
                frame, camera_pose, _, _ = tracker.process_frame(frame)
                cam_traj_writer.write(cap.get(cv2.CAP_PROP_POS_MSEC), camera_pose)
            cap.release()
    tracker.save_map(str(out_dir / "map.txt"))


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Run the SLAM system")
    parser.add_argument("--video", type=str, help="Path to video file", default="input_video.mp4")
    parser.add_argument("--filter", type=str, help="Filter to use (ekf)", default="ekf")
    parser.add_argument("--detections", type=str, default=None,
                        help="replay file with pre-computed ArUco detections (.npz)")
    parser.add_argument("--output-dir", dest="output_dir", type=str, default="outputs")
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
