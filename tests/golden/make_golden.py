#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run in the build container only (``/root/reference`` is not on the GPU box):

    python tests/golden/make_golden.py

The reference is imported as it lies under /root/reference (nothing is copied):
``cv2`` and ``gtsam`` are absent from the image and are replaced by
``MagicMock`` modules (the EKF never touches them after the detector is
constructed, base_filter.py:74-90); the ``aruco_slam.`` import prefix used by
extended_kalman_filter.py:12 is satisfied by a symlink in /tmp; the CWD is the
reference root because BaseFilter loads ./calibration/*.npy
(base_filter.py:12-13,55-63).

Fixtures written (inputs + the reference's outputs, data only):

  g1_measurement_model.npz   h / dh lambdas on 256 random 13-vectors
  g2_teacher_forced.npz      (state, P, ids, poses) -> (state', P') single steps
  g3_free_run.npz/.txt       200-frame <=10-marker run, trajectory.txt, map.txt
  g4_scale_n256.npz          n=256,  m=16: bootstrap + 5 frames, checksums
  g4_scale_n1024.npz         n=1024, m=32: bootstrap + 3 frames, checksums
  g6_quaternion_rule.npz     SciPy's behaviour at the call site :138-149
  c1_detections.npz          the synthetic C1 replay itself (inputs only)
  g5_trajectory.txt, g5_map.txt, g5_detections.npz
                             the same EKF_Rotations run driven like main/run_slam.py (``g5txt`` mode)
  g5_rotations.npz           EKF_Rotations (ekf_with_rotations.py): h/dh lambdas on 128 random
                             20-vectors, 120-frame 6-marker free run with full (state, P)
                             snapshots before/after 10 of its steps
"""
from __future__ import annotations

import os
import sys
import tempfile
import time
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")

sys.path.insert(0, str(REPO))
from aruco_slam_amd.synthetic import SyntheticStream, small_sequence  # noqa: E402


def load_reference():
    for name in ("cv2", "cv2.aruco", "gtsam", "gtsam.symbol_shorthand"):
        sys.modules[name] = MagicMock()
    link_dir = Path(tempfile.mkdtemp(prefix="oracle_ref_"))
    (link_dir / "aruco_slam").symlink_to(REF)
    sys.path.insert(0, str(link_dir))
    sys.path.insert(0, str(REF))
    os.chdir(REF)
    from aruco_slam.filters.extended_kalman_filter import EKF
    from outputs.trajectory_writer import TrajectoryWriter
    return EKF, TrajectoryWriter


def golden_rotations():
    """G5: the reference's EKF_Rotations (it caches its SymPy lambdas with dill under /tmp)."""
    from filters.ekf_with_rotations import EKF_Rotations
    init_pose = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
    t0 = time.time()
    flt = EKF_Rotations(init_pose)
    print(f"reference EKF_Rotations ctor {time.time() - t0:.1f}s")
    rng = np.random.default_rng(4321)
    x = rng.normal(0.0, 1.0, size=(128, 20))
    x[:, 7:10] = 0.0
    x[:, 17:20] = 0.0
    x[:64, 3:7] /= np.linalg.norm(x[:64, 3:7], axis=1, keepdims=True)
    x[:64, 13:17] /= np.linalg.norm(x[:64, 13:17], axis=1, keepdims=True)
    x[:, 10:13] = x[:, 0:3] + rng.normal(0.0, 5.0, size=(128, 3))
    hv = np.stack([np.asarray(flt.h(list(r)), dtype=np.float64).reshape(7) for r in x])
    dh = np.stack([np.asarray(flt.partial_jacobian(list(r)), dtype=np.float64) for r in x])
    out = {"x": x, "h": hv, "dh": dh}
    seq = small_sequence(frames=120, markers=6, max_visible=4, seed=3)
    cat_ids, cat_poses, offs, has = [], [], [0], []
    snaps = {1, 2, 5, 14, 15, 30, 47, 48, 90, 119}
    hist = []
    for f, (ts, ids, poses) in enumerate(seq):
        has.append(ids is not None)
        if ids is not None:
            cat_ids.append(ids)
            cat_poses.append(poses)
            offs.append(offs[-1] + len(ids))
            if f in snaps:
                out[f"f{f}_state0"] = np.asarray(flt.state, dtype=np.float64).copy()
                out[f"f{f}_P0"] = np.asarray(flt.uncertainty, dtype=np.float64).copy()
                out[f"f{f}_lm_ids"] = np.asarray(
                    [k for k, _ in sorted(flt.landmarks.items(), key=lambda kv: kv[1])], dtype=np.int64)
            flt.observe(ids, poses)
            if f in snaps:
                out[f"f{f}_state1"] = np.asarray(flt.state, dtype=np.float64).copy()
                out[f"f{f}_P1"] = np.asarray(flt.uncertainty, dtype=np.float64).copy()
        else:
            offs.append(offs[-1])
        cam = np.zeros(10)
        cam[:] = np.asarray(flt.state[:10], dtype=np.float64)
        hist.append(cam[:7].copy())
    out.update(frames=np.asarray(sorted(snaps)), ids=np.concatenate(cat_ids).astype(np.int32),
               poses=np.concatenate(cat_poses), offsets=np.asarray(offs, dtype=np.int64),
               has_detections=np.asarray(has), cam=np.stack(hist),
               final_state=np.asarray(flt.state, dtype=np.float64),
               final_P=np.asarray(flt.uncertainty, dtype=np.float64),
               lm_ids=np.asarray([k for k, _ in sorted(flt.landmarks.items(), key=lambda kv: kv[1])]))
    np.savez_compressed(HERE / "g5_rotations.npz", **out)
    print("g5 done")


def golden_rotations_app_loop(traj_writer_cls):
    """G5 text outputs: the SAME 120-frame sequence as g5_rotations.npz driven the way
    main/run_slam.py:110-143 drives a tracker (observe only on frames with detections, get_poses
    every frame, TrajectoryWriter line per frame, save_map at the end) with the reference's
    EKF_Rotations -> g5_trajectory.txt, g5_map.txt, and the replay inputs g5_detections.npz."""
    from filters.ekf_with_rotations import EKF_Rotations
    flt = EKF_Rotations(np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0]))
    seq = small_sequence(frames=120, markers=6, max_visible=4, seed=3)
    tmpdir = Path(tempfile.mkdtemp(prefix="g5txt_"))
    cat_ids, cat_poses, offs, has, stamps = [], [], [0], [], []
    with traj_writer_cls(str(tmpdir / "trajectory.txt")) as writer:
        for ts, ids, poses in seq:
            stamps.append(ts)
            has.append(ids is not None)
            if ids is not None:                            # base_filter.py:197-204
                flt.observe(ids, poses)
                cat_ids.append(ids)
                cat_poses.append(poses)
                offs.append(offs[-1] + len(ids))
            else:
                offs.append(offs[-1])
            camera_pose, _ = flt.get_poses()               # base_filter.py:206-207
            writer.write(ts, camera_pose)                  # run_slam.py:124-125
    flt.save_map(str(tmpdir / "map.txt"))                  # run_slam.py:143
    (HERE / "g5_trajectory.txt").write_text((tmpdir / "trajectory.txt").read_text())
    (HERE / "g5_map.txt").write_text((tmpdir / "map.txt").read_text())
    np.savez_compressed(HERE / "g5_detections.npz", ids=np.concatenate(cat_ids).astype(np.int32),
                        poses=np.concatenate(cat_poses), offsets=np.asarray(offs, dtype=np.int64),
                        timestamps_ms=np.asarray(stamps, dtype=np.float64), has_detections=np.asarray(has))
    g5 = np.load(HERE / "g5_rotations.npz", allow_pickle=False)
    assert np.array_equal(g5["final_state"], np.asarray(flt.state, dtype=np.float64)), "not the g5 run"
    print("g5 app-loop outputs done")


# Seed of the C1 replay.  The as-written reference is chaotic: a
# 1e-15 input perturbation reaches 1e-4 within 100-200 frames for every seed
# tried (0..15); seed 6 has the longest horizon.  The fixture therefore also
# stores a second reference run on inputs perturbed by 1e-15 (relative): the
# divergence between the two IS the reference's own reproducibility envelope
# and defines the free-run comparison horizon used by the tests.
C1_SEED = 6


def golden_calibration():
    """The reference's own camera calibration (calibration/camera_matrix.npy, dist_coeffs.npy: plain numeric
    arrays, loaded without pickle) as a fixture for the detection -> pose tests."""
    ref = Path("/root/reference/calibration")
    k = np.load(ref / "camera_matrix.npy", allow_pickle=False)
    d = np.load(ref / "dist_coeffs.npy", allow_pickle=False)
    np.savez(HERE / "calibration.npz", camera_matrix=k, dist_coeffs=d)
    print("calibration.npz", k.shape, d.shape)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "calib":
        golden_calibration()
        return
    ekf_cls, traj_writer_cls = load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "g5":
        golden_rotations()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g5txt":
        golden_rotations_app_loop(traj_writer_cls)
        return
    init_pose = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])   # main/run_slam.py:85-88

    t0 = time.time()
    first = ekf_cls(init_pose)
    print(f"reference ctor (SymPy) {time.time() - t0:.1f}s")
    lambdas = (first.h, first.partial_jacobian)
    ekf_cls.initialize_h = lambda self: lambdas     # in-process reuse only

    rng = np.random.default_rng(1234)

    # ---- G1: measurement model -------------------------------------------
    x = rng.normal(0.0, 1.0, size=(256, 13))
    x[:, 7:10] = 0.0                                 # e = 0 at every call site (:152)
    x[:128, 3:7] /= np.linalg.norm(x[:128, 3:7], axis=1, keepdims=True)
    x[128:, 3:7] *= rng.uniform(0.5, 2.0, size=(128, 1))
    x[:, 10:13] = x[:, 0:3] + rng.normal(0.0, 5.0, size=(256, 3))
    hv = np.stack([np.asarray(first.h(list(r)), dtype=np.float64).reshape(3) for r in x])
    dh = np.stack([np.asarray(first.partial_jacobian(list(r)), dtype=np.float64) for r in x])
    np.savez_compressed(HERE / "g1_measurement_model.npz", x=x, h=hv, dh=dh)
    print("g1 done")

    # ---- G3 free-run + G2 teacher-forced snapshots -------------------------
    seq = small_sequence(frames=200, markers=10, max_visible=6, seed=C1_SEED)
    cat_ids, cat_poses, offs, stamps, has = [], [], [0], [], []
    for ts, ids, poses in seq:
        stamps.append(ts)
        has.append(ids is not None)
        if ids is not None:
            cat_ids.append(ids)
            cat_poses.append(poses)
            offs.append(offs[-1] + len(ids))
        else:
            offs.append(offs[-1])
    np.savez_compressed(HERE / "c1_detections.npz",
                        ids=np.concatenate(cat_ids).astype(np.int32),
                        poses=np.concatenate(cat_poses),
                        offsets=np.asarray(offs, dtype=np.int64),
                        timestamps_ms=np.asarray(stamps),
                        has_detections=np.asarray(has))

    ekf = ekf_cls(init_pose)
    snap_frames = {1, 2, 3, 7, 12, 13, 24, 48, 49, 60, 90, 120, 150, 199}
    g2 = {}
    cam_hist = np.zeros((len(seq), 7))
    tmpdir = Path(tempfile.mkdtemp(prefix="golden_out_"))
    with traj_writer_cls(str(tmpdir / "trajectory.txt")) as writer:
        for f, (ts, ids, poses) in enumerate(seq):
            if ids is not None:
                if f in snap_frames:
                    g2[f"f{f}_state0"] = np.asarray(ekf.state, dtype=np.float64).copy()
                    g2[f"f{f}_P0"] = np.asarray(ekf.uncertainty, dtype=np.float64).copy()
                    g2[f"f{f}_lm_ids"] = np.asarray(
                        [k for k, _ in sorted(ekf.landmarks.items(), key=lambda kv: kv[1])],
                        dtype=np.int64)
                    g2[f"f{f}_ids"] = np.asarray(ids)
                    g2[f"f{f}_poses"] = np.asarray(poses)
                ekf.observe(ids, poses)              # boundary: base_filter.py:203-204
                if f in snap_frames:
                    g2[f"f{f}_state1"] = np.asarray(ekf.state, dtype=np.float64).copy()
                    g2[f"f{f}_P1"] = np.asarray(ekf.uncertainty, dtype=np.float64).copy()
            cam, _ = ekf.get_poses()                 # base_filter.py:206-207
            writer.write(ts, cam)                    # main/run_slam.py:124-125
            cam_hist[f] = np.asarray(cam[:7], dtype=np.float64)
    ekf.save_map(str(tmpdir / "map.txt"))            # main/run_slam.py:143
    # second reference run, inputs perturbed by 1e-15 relative
    prng = np.random.default_rng(7)
    ekf_p = ekf_cls(init_pose)
    cam_pert = np.zeros((len(seq), 7))
    for f, (ts, ids, poses) in enumerate(seq):
        if ids is not None:
            ekf_p.observe(ids, poses * (1.0 + 1e-15 * prng.choice([-1.0, 1.0], size=poses.shape)))
        cam_pert[f] = np.asarray(ekf_p.get_poses()[0][:7], dtype=np.float64)
    g2["frames"] = np.asarray(sorted(snap_frames))
    np.savez_compressed(HERE / "g2_teacher_forced.npz", **g2)
    np.savez_compressed(
        HERE / "g3_free_run.npz", cam=cam_hist, cam_perturbed=cam_pert,
        final_state=np.asarray(ekf.state, dtype=np.float64),
        final_P=np.asarray(ekf.uncertainty, dtype=np.float64),
        lm_ids=np.asarray([k for k, _ in sorted(ekf.landmarks.items(), key=lambda kv: kv[1])]))
    (HERE / "g3_trajectory.txt").write_text((tmpdir / "trajectory.txt").read_text())
    (HERE / "g3_map.txt").write_text((tmpdir / "map.txt").read_text())
    print("g2/g3 done")

    # ---- G4: scale ----------------------------------------------------------
    for n, m, steady in ((256, 16, 5), (1024, 32, 3)):
        t0 = time.time()
        stream = SyntheticStream(n, m, seed=0)
        ekf = ekf_cls(init_pose)
        all_ids, all_z = [], []
        for ids, poses in stream.bootstrap():
            ekf.observe(ids, poses)
            all_ids.append(ids)
            all_z.append(poses[:, 0:3])
        boot_state = np.asarray(ekf.state, dtype=np.float64).copy()
        boot_diag = np.diagonal(np.asarray(ekf.uncertainty)).copy()
        states, diags, fro, asym, blocks = [], [], [], [], []
        brng = np.random.default_rng(99)
        n_dims = 3 * n + 10
        corners = brng.integers(0, n_dims - 16, size=(8, 2))
        corners[0] = (0, 0)
        for ids, poses in stream.steady(steady):
            ekf.observe(ids, poses)
            all_ids.append(ids)
            all_z.append(poses[:, 0:3])
            p = np.asarray(ekf.uncertainty, dtype=np.float64)
            states.append(np.asarray(ekf.state, dtype=np.float64).copy())
            diags.append(np.diagonal(p).copy())
            fro.append(np.linalg.norm(p))
            asym.append(np.abs(p - p.T).max())
            blocks.append(np.stack([p[r:r + 16, c:c + 16] for r, c in corners]))
        np.savez_compressed(
            HERE / f"g4_scale_n{n}.npz", n=n, m=m, seed=0,
            ids=np.stack(all_ids), z=np.stack(all_z),
            boot_frames=stream.bootstrap_frames,
            boot_state=boot_state, boot_diag=boot_diag,
            states=np.stack(states), diags=np.stack(diags),
            fro=np.asarray(fro), asym=np.asarray(asym),
            block_corners=corners, blocks=np.stack(blocks))
        print(f"g4 n={n} done in {time.time() - t0:.1f}s")

    # ---- G6: quaternion rule at the call site ------------------------------
    from scipy.spatial.transform import Rotation
    qs = rng.normal(size=(64, 4))
    qs[:32] /= np.linalg.norm(qs[:32], axis=1, keepdims=True)
    errs = rng.normal(0.0, 0.2, size=(64, 3))
    outs = np.zeros((64, 4))
    for i in range(64):
        # same calls as extended_kalman_filter.py:138-149
        q = Rotation.from_quat(qs[i])
        dq = Rotation.from_quat([1, *errs[i] / 2])
        outs[i] = (dq * q).as_quat(scalar_first=True)
    np.savez_compressed(HERE / "g6_quaternion_rule.npz", q=qs, err=errs, out=outs)
    print("g6 done")
    golden_rotations()


if __name__ == "__main__":
    main()
