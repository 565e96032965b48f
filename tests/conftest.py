import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / "tests" / "golden"
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_npz(name):
    return np.load(GOLDEN / name, allow_pickle=False)


def rel_err(a, b):
    """max |a-b| / max(1, max|b|): the '<=1e-4 relative' measure used for
    trajectory / map numbers (BASELINE.json north_star)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def rel_err_elem(a, b, floor=1e-3):
    """Element-wise relative error max_i |a_i - b_i| / max(|b_i|, floor).  Most trajectory / map /
    covariance entries are < 1, so ``rel_err`` (normalised by max(1, max|b|)) is an ABSOLUTE error
    on them; this one is relative for every entry larger than ``floor`` and absolute (in units of
    ``floor``) below it."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def report(test, **values):
    """Measured parity numbers of the GPU tests, one JSON line each, into
    ``gpurun_out/parity_metrics.jsonl`` (scratch; the summary that is judged is copied to
    ``profiles/``) and onto stdout (``pytest -s``)."""
    import json
    line = json.dumps({"test": test, **{k: (float(v) if isinstance(v, (int, float, np.floating)) else v)
                                        for k, v in values.items()}})
    print("PARITY", line)
    out = REPO / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        with (out / "parity_metrics.jsonl").open("a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass


def chaos_horizon(free_run, envelope=1e-8):
    """Last frame up to which the reference reproduces ITSELF: the fixture
    holds a second reference run on inputs perturbed by 1e-15 (relative); the
    as-written filter is chaotic and that perturbation grows to
    O(1) within 200 frames.  Free-run comparisons are made up to the last
    frame where the two reference runs still agree to ``envelope``."""
    d = np.abs(free_run["cam"] - free_run["cam_perturbed"]).max(axis=1)
    bad = np.nonzero(d > envelope)[0]
    return int(bad[0]) - 1 if len(bad) else len(d) - 1


def synthetic_marker_views(count, seed=0, marker_size=0.16, width=1920, height=1080):
    """Random marker poses in front of the reference's calibrated camera (tests/golden/calibration.npz) whose four
    corners all fall inside the image: (camera_matrix, dist, corners_px [count,4,2], tvecs [count,3], rotations)."""
    import numpy as np
    from pathlib import Path
    from scipy.spatial.transform import Rotation
    from oracle.ippe_numpy import object_points, project_points
    cal = np.load(Path(__file__).resolve().parent / "golden" / "calibration.npz", allow_pickle=False)
    k, dist = cal["camera_matrix"], cal["dist_coeffs"].reshape(-1)
    rng = np.random.default_rng(seed)
    corners, tvecs, rots = [], [], []
    while len(corners) < count:
        rvec = rng.normal(size=3)
        rvec *= rng.uniform(0.0, 1.2) / np.linalg.norm(rvec)
        rot = Rotation.from_rotvec(rvec) * Rotation.from_euler("x", np.pi)      # marker faces the camera
        t = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-0.8, 0.8), rng.uniform(0.4, 4.0)])
        px = project_points(object_points(marker_size) @ rot.as_matrix().T + t, k, dist)
        if px[:, 0].min() < 0 or px[:, 0].max() > width or px[:, 1].min() < 0 or px[:, 1].max() > height:
            continue
        corners.append(px); tvecs.append(t); rots.append(rot)
    return k, dist, np.stack(corners), np.stack(tvecs), rots
