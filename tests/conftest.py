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


def chaos_horizon(free_run, envelope=1e-8):
    """Last frame up to which the reference reproduces ITSELF: the fixture
    holds a second reference run on inputs perturbed by 1e-15 (relative); the
    as-written filter is chaotic and that perturbation grows to
    O(1) within 200 frames.  Free-run comparisons are made up to the last
    frame where the two reference runs still agree to ``envelope``."""
    d = np.abs(free_run["cam"] - free_run["cam_perturbed"]).max(axis=1)
    bad = np.nonzero(d > envelope)[0]
    return int(bad[0]) - 1 if len(bad) else len(d) - 1
