"""Pin the CPU oracle (oracle/ekf_numpy.py) against fixtures produced by the
real reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import chaos_horizon, load_npz, rel_err
from oracle import ekf_numpy as orc

INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])


def test_g1_measurement_model_closed_form():
    g = load_npz("g1_measurement_model.npz")
    for x, h, dh in zip(g["x"], g["h"], g["dh"]):
        scale = max(1.0, np.abs(dh).max())
        assert np.abs(orc.h_closed(x) - h).max() <= 1e-12 * max(1.0, np.abs(h).max())
        assert np.abs(orc.dh_closed(x) - dh).max() <= 1e-12 * scale


def test_g6_quaternion_rule_as_written():
    g = load_npz("g6_quaternion_rule.npz")
    for q, e, out in zip(g["q"], g["err"], g["out"]):
        assert np.abs(orc.quat_update_as_written(q, e) - out).max() <= 1e-15 * 8


def _restore(flt, state, cov, lm_ids):
    flt.state = state.copy()
    flt.uncertainty = cov.copy()
    flt.landmarks = {int(k): i for i, k in enumerate(lm_ids)}
    flt.num_landmarks = len(lm_ids)


@pytest.mark.parametrize("mode", ["reference_ops", "fast"])
def test_g2_teacher_forced_steps(mode):
    g = load_npz("g2_teacher_forced.npz")
    for f in g["frames"]:
        flt = orc.OracleEKF(INIT, mode=mode)
        _restore(flt, g[f"f{f}_state0"], g[f"f{f}_P0"], g[f"f{f}_lm_ids"])
        flt.observe(list(g[f"f{f}_ids"]), g[f"f{f}_poses"])
        assert flt.state.shape == g[f"f{f}_state1"].shape
        assert rel_err(flt.state, g[f"f{f}_state1"]) <= 1e-11, f
        assert rel_err(flt.uncertainty, g[f"f{f}_P1"]) <= 1e-11, f


def _replay(flt):
    det = load_npz("c1_detections.npz")
    offs = det["offsets"]
    cams = []
    for f in range(len(det["timestamps_ms"])):
        if det["has_detections"][f]:
            sl = slice(offs[f], offs[f + 1])
            flt.observe(list(det["ids"][sl]), det["poses"][sl])
        cams.append(np.asarray(flt.get_poses()[0][:7], dtype=np.float64).copy())
    return np.stack(cams)


@pytest.mark.parametrize("mode", ["reference_ops", "fast"])
def test_g3_free_run_200_frames(mode):
    g = load_npz("g3_free_run.npz")
    flt = orc.OracleEKF(INIT, mode=mode)
    cams = _replay(flt)
    # chaotic reference: compare inside the horizon over which the
    # reference reproduces itself under a 1e-15 input perturbation (amplified
    # <= 1e7 there, so 1e-16 rounding differences stay <= ~1e-8)
    hz = chaos_horizon(g)
    assert hz >= 120
    assert rel_err(cams[:hz + 1], g["cam"][:hz + 1]) <= 1e-6
    # beyond the horizon only boundedness is meaningful
    assert np.isfinite(cams).all() and np.abs(cams).max() < 50.0
    assert [k for k, _ in sorted(flt.landmarks.items(), key=lambda kv: kv[1])] == list(g["lm_ids"])


@pytest.mark.parametrize("name,mode", [("g4_scale_n256.npz", "reference_ops"),
                                       ("g4_scale_n256.npz", "fast"),
                                       ("g4_scale_n1024.npz", "fast")])
def test_g4_scale(name, mode):
    g = load_npz(name)
    flt = orc.OracleEKF(INIT, mode=mode)
    boot = int(g["boot_frames"])
    for f in range(boot):
        poses = np.zeros((g["ids"].shape[1], 6))
        poses[:, :3] = g["z"][f]
        flt.observe(list(g["ids"][f]), poses)
    assert rel_err(flt.state, g["boot_state"]) <= 1e-8
    assert rel_err(np.diagonal(flt.uncertainty), g["boot_diag"]) <= 1e-8
    for j in range(g["states"].shape[0]):
        poses = np.zeros((g["ids"].shape[1], 6))
        poses[:, :3] = g["z"][boot + j]
        flt.observe(list(g["ids"][boot + j]), poses)
        p = np.asarray(flt.uncertainty)
        assert rel_err(flt.state, g["states"][j]) <= 1e-8
        assert rel_err(np.diagonal(p), g["diags"][j]) <= 1e-8
        assert abs(np.linalg.norm(p) - g["fro"][j]) <= 1e-8 * g["fro"][j]
        for (r, c), blk in zip(g["block_corners"], g["blocks"][j]):
            assert rel_err(p[r:r + 16, c:c + 16], blk) <= 1e-8


def test_synthetic_stream_matches_fixture_inputs():
    """The seeded generator reproduces the inputs the fixture was made with."""
    from aruco_slam_amd.synthetic import SyntheticStream
    g = load_npz("g4_scale_n256.npz")
    s = SyntheticStream(256, 16, seed=0)
    f = 0
    for ids, poses in s.bootstrap():
        assert np.array_equal(ids, g["ids"][f])
        assert np.allclose(poses[:, :3], g["z"][f], rtol=0, atol=1e-12)
        f += 1
    for ids, poses in s.steady(5):
        assert np.array_equal(ids, g["ids"][f])
        assert np.allclose(poses[:, :3], g["z"][f], rtol=0, atol=1e-12)
        f += 1


# ---------------------------------------------------------------------------
# EKF_Rotations (ekf_with_rotations.py): G5
# ---------------------------------------------------------------------------
def test_g5_rotations_measurement_model_closed_form():
    g = load_npz("g5_rotations.npz")
    for x, h, dh in zip(g["x"], g["h"], g["dh"]):
        assert np.abs(orc.h_rot_closed(x) - h).max() <= 1e-12 * max(1.0, np.abs(h).max())
        assert np.abs(orc.dh_rot_closed(x) - dh).max() <= 1e-12 * max(1.0, np.abs(dh).max())


@pytest.mark.parametrize("mode", ["reference_ops", "fast"])
def test_g5_rotations_teacher_forced_steps(mode):
    g = load_npz("g5_rotations.npz")
    offs = g["offsets"]
    for f in g["frames"]:
        flt = orc.OracleEKFRotations(INIT, mode=mode)
        _restore(flt, g[f"f{f}_state0"], g[f"f{f}_P0"], g[f"f{f}_lm_ids"])
        sl = slice(offs[f], offs[f + 1])
        flt.observe(list(g["ids"][sl]), g["poses"][sl])
        assert flt.state.shape == g[f"f{f}_state1"].shape
        assert rel_err(flt.state, g[f"f{f}_state1"]) <= 1e-11, f
        assert rel_err(flt.uncertainty, g[f"f{f}_P1"]) <= 1e-11, f


@pytest.mark.parametrize("mode", ["reference_ops", "fast"])
def test_g5_rotations_free_run_120_frames(mode):
    """This variant uses the consistent quaternion convention: not chaotic, the whole run matches."""
    g = load_npz("g5_rotations.npz")
    offs = g["offsets"]
    flt = orc.OracleEKFRotations(INIT, mode=mode)
    cams = []
    for f in range(len(g["has_detections"])):
        if g["has_detections"][f]:
            sl = slice(offs[f], offs[f + 1])
            flt.observe(list(g["ids"][sl]), g["poses"][sl])
        cams.append(np.asarray(flt.state[:7], dtype=np.float64).copy())
    assert rel_err(np.stack(cams), g["cam"]) <= 1e-9
    assert rel_err(flt.state, g["final_state"]) <= 1e-9
    assert rel_err(flt.uncertainty, g["final_P"]) <= 1e-9


def test_g5_rotations_app_loop_text_outputs(golden_dir):
    """The oracle driven like main/run_slam.py reproduces the numbers of the reference's
    EKF_Rotations trajectory.txt / map.txt (g5txt fixtures) over all 120 frames."""
    from oracle.ekf_numpy import OracleEKFRotations
    det = load_npz("g5_detections.npz")
    offs = det["offsets"]
    orc = OracleEKFRotations(INIT, mode="fast")
    rows = []
    for f in range(len(det["timestamps_ms"])):
        if det["has_detections"][f]:
            sl = slice(int(offs[f]), int(offs[f + 1]))
            orc.observe(list(det["ids"][sl]), det["poses"][sl])
        rows.append(np.asarray(orc.state[:7], dtype=np.float64))
    ref = np.array([[float(t) for t in ln.split()] for ln in (golden_dir / "g5_trajectory.txt").read_text().splitlines()])
    assert ref.shape == (120, 8)
    assert np.abs(ref[:, 0] - np.round(det["timestamps_ms"] / 1000.0, 4)).max() <= 1e-12
    assert rel_err(np.stack(rows), ref[:, 1:]) <= 1e-9
    lines = (golden_dir / "g5_map.txt").read_text().splitlines()[4:]
    ids = [int(lines[i]) for i in range(0, len(lines) - 2, 4)]
    pose = np.array([[float(t) for t in lines[i + 1].split(", ")] for i in range(0, len(lines) - 2, 4)])
    unc = np.array([[float(t) for t in lines[i + 2].split(", ")] for i in range(0, len(lines) - 2, 4)])
    assert ids == [k for k, _ in sorted(orc.landmarks.items(), key=lambda kv: kv[1])]
    assert rel_err(pose, orc.get_poses()[1]) <= 1e-9
    assert rel_err(unc, orc.get_lm_uncertainties()) <= 1e-9
