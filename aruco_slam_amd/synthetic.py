"""Deterministic synthetic ArUco detections (no video / cv2 in this image).

Produces exactly what the vision front-end hands to the filter boundary
(reference ``BaseFilter.process_frame`` -> ``self.observe(ids, poses)``,
filters/base_filter.py:194-204): per frame a list of marker ids and an
``(m, 6)`` float64 array ``[tvec | rvec]`` in the camera frame.  Only
``pose[0:3]`` is used by the EKF (extended_kalman_filter.py:192,196,272).

Stream definition: landmarks ``U([-10,10]^2 x [5,25])``,
camera ``c(t) = (0.5 sin t, 0.2 sin 2t, 0.05 t)``,
``R(t) = Rz(0.05 t) Ry(0.15 sin 0.7t) Rx(0.1 sin t)``, ``t = frame / 30``;
bootstrap frames show ids ``j*m .. j*m+m-1``; steady-state frames show a sorted
random subset of m ids with ``z = R(t)^T (l - c(t)) + N(0, 0.01^2)``.
"""
from __future__ import annotations

import numpy as np


def camera_truth(frame: int):
    t = frame / 30.0
    c = np.array([0.5 * np.sin(t), 0.2 * np.sin(2.0 * t), 0.05 * t])
    ax, ay, az = 0.1 * np.sin(t), 0.15 * np.sin(0.7 * t), 0.05 * t
    cx, sx = np.cos(ax), np.sin(ax)
    cy, sy = np.cos(ay), np.sin(ay)
    cz, sz = np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return c, rz @ ry @ rx


class SyntheticStream:
    """n landmarks, m visible per frame, seeded."""

    def __init__(self, n: int, m: int, seed: int = 0, noise: float = 0.01, rvec_sigma: float = 0.0):
        if m > n:
            raise ValueError("m must be <= n")
        self.n, self.m, self.noise = n, m, noise
        self.rng = np.random.default_rng(seed)
        lm = np.empty((n, 3))
        lm[:, 0:2] = self.rng.uniform(-10.0, 10.0, size=(n, 2))
        lm[:, 2] = self.rng.uniform(5.0, 25.0, size=n)
        self.landmarks = lm
        self.rvec_sigma = rvec_sigma     # > 0: poses[:, 3:6] carry noisy marker orientations (EKF_Rotations)
        self.frame = 0

    @property
    def bootstrap_frames(self) -> int:
        return -(-self.n // self.m)

    def _observe(self, ids):
        c, rot = camera_truth(self.frame)
        z = (self.landmarks[ids] - c) @ rot          # rows = R^T (l - c)
        z = z + self.rng.normal(0.0, self.noise, size=z.shape)
        poses = np.zeros((len(ids), 6))
        poses[:, 0:3] = z
        if self.rvec_sigma > 0.0:
            poses[:, 3:6] = self.rng.normal(0.0, self.rvec_sigma, size=(len(ids), 3))
        self.frame += 1
        return np.asarray(ids, dtype=np.int32), poses

    def bootstrap(self):
        """Frames that introduce every landmark through ``observe()`` only."""
        for j in range(self.bootstrap_frames):
            lo = j * self.m
            ids = np.arange(lo, min(lo + self.m, self.n))
            yield self._observe(ids)

    def steady(self, frames: int):
        for _ in range(frames):
            ids = np.sort(self.rng.choice(self.n, self.m, replace=False))
            yield self._observe(ids)


def small_sequence(frames: int = 200, markers: int = 10, max_visible: int = 6,
                   seed: int = 0):
    """C1-sized replay: <=10 markers, 1..max_visible seen per frame, markers
    appear progressively, some frames empty, occasional duplicate ids
    (legal at the boundary: SURVEY 8(a) a2).  Returns a list of
    ``(timestamp_ms, ids, poses)``; ids is ``None`` for frames without
    detections (base_filter.py:197)."""
    rng = np.random.default_rng(seed)
    lm = np.empty((markers, 3))
    lm[:, 0:2] = rng.uniform(-2.0, 2.0, size=(markers, 2))
    lm[:, 2] = rng.uniform(3.0, 8.0, size=markers)
    marker_ids = rng.permutation(50)[:markers]       # DICT_5X5_50 id range
    out = []
    for f in range(frames):
        ts = (f + 1) * 1000.0 / 30.0
        if f in (0, 57, 58, 140):                     # frames with no detections
            out.append((ts, None, np.array([])))
            continue
        known = min(markers, 2 + f // 12)
        m = int(rng.integers(1, min(max_visible, known) + 1))
        pick = np.sort(rng.choice(known, m, replace=False))
        if f % 41 == 7 and m >= 2:                    # duplicate detection
            pick[-1] = pick[0]
        c, rot = camera_truth(f)
        z = (lm[pick] - c) @ rot + rng.normal(0.0, 0.01, size=(m, 3))
        poses = np.zeros((m, 6))
        poses[:, 0:3] = z
        poses[:, 3:6] = rng.normal(0.0, 0.05, size=(m, 3))
        out.append((ts, marker_ids[pick].astype(np.int32), poses))
    return out
