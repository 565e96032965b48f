"""CPU-only tests: the C-ABI library loads and exports what the header
declares, host logic, file formats, synthetic streams.  No GPU compute."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import REPO, load_npz


@pytest.fixture(scope="module")
def lib():
    from aruco_slam_amd import _build, hip_backend
    _build.build()
    return hip_backend.load_library()


def test_header_symbols_are_exported(lib):
    from aruco_slam_amd import hip_backend
    header = (REPO / "include" / "ekf_slam_hip.h").read_text()
    declared = set(re.findall(r"\b(ekf_[a-z_0-9]+)\s*\(", header))
    assert declared == set(hip_backend.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_config_struct_matches_header_and_defaults(lib):
    from aruco_slam_amd.hip_backend import EkfConfig
    cfg = EkfConfig()
    assert lib.ekf_default_config(ctypes.byref(cfg)) == 0
    # extended_kalman_filter.py:21-27
    assert (cfg.initial_camera_uncertainty, cfg.initial_landmark_uncertainty) == (0.1, 0.7)
    assert (cfg.r_uncertainty, cfg.q_cam, cfg.q_err, cfg.q_lm) == (0.9, 0.3, 0.5, 0.01)
    assert ctypes.sizeof(EkfConfig) == 8 * 4 + 6 * 8 + 8


@pytest.mark.parametrize("n,m,dtype,elem", [(1024, 32, 1, 4), (256, 16, 0, 8), (4096, 64, 1, 4)])
def test_query_sizes(lib, n, m, dtype, elem):
    from aruco_slam_amd.hip_backend import EkfConfig
    cfg = EkfConfig()
    lib.ekf_default_config(ctypes.byref(cfg))
    cfg.max_landmarks, cfg.max_visible, cfg.cov_dtype = n, m, dtype
    ld, cb, sb, wb = ctypes.c_int64(), ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.ekf_query_sizes(ctypes.byref(cfg), ctypes.byref(ld), ctypes.byref(cb),
                               ctypes.byref(sb), ctypes.byref(wb)) == 0
    dims = 3 * n + 10
    assert ld.value % 128 == 0 and dims <= ld.value < dims + 128
    assert cb.value == ld.value * ld.value * elem
    assert sb.value == ld.value * 8
    assert wb.value > 3 * m * ld.value * 8


def test_bad_config_is_rejected_with_message(lib):
    from aruco_slam_amd.hip_backend import EkfConfig
    cfg = EkfConfig()
    lib.ekf_default_config(ctypes.byref(cfg))
    cfg.max_visible = 65
    assert lib.ekf_query_sizes(ctypes.byref(cfg), None, None, None, None) == -1
    assert b"max_visible" in lib.ekf_last_error_string()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from aruco_slam_amd.filters.extended_kalman_filter import EKF
    with pytest.raises(RuntimeError):
        EKF(np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0]))


def test_product_never_imports_the_oracle():
    for py in (REPO / "aruco_slam_amd").rglob("*.py"):
        text = py.read_text()
        assert "oracle" not in text.replace("# oracle", ""), py


class _StubFilter:
    """BaseFilter with a fixed map, to exercise the file formats on CPU."""

    def __new__(cls, *a, **k):
        from aruco_slam_amd.filters.base_filter import BaseFilter

        class Stub(BaseFilter):
            def __init__(self):
                super().__init__(np.zeros(10), None)
                self.landmarks = {}
                self.pos, self.unc = [], []

            def add_marker(self, idx, pose, uncertainity=None):
                self.landmarks[idx] = len(self.pos)
                self.pos.append(np.asarray(pose, dtype=np.float64)[:3])
                self.unc.append(np.asarray(uncertainity, dtype=np.float64))

            def get_poses(self):
                return np.zeros(10), np.asarray(self.pos).reshape(-1, 3)

            def get_lm_estimates(self):
                return self.landmarks.items()

            def get_lm_uncertainties(self):
                return np.asarray(self.unc).reshape(-1, 3)

        return Stub()


def test_save_map_format_and_load_map_round_trip(tmp_path):
    a = _StubFilter()
    a.add_marker(10, [1.1191539191370652, 0.523044137695957, 4.781684421973866],
                 [1.318541059042735, 1.7085071611126854, 1.4905500967390102])
    a.add_marker(8, [-1.5711137587719017, 0.050712492645848636, 5.602895975893089],
                 [1.1228681750268494, 1.6589399303511667, 1.5492904185599448])
    out = tmp_path / "map.txt"
    a.save_map(str(out))
    text = out.read_text()
    # byte-identical to the first two entries of the reference's sample map
    # (/root/reference/outputs/map.txt:1-12, format fixture)
    assert text == ("# landmark_id\n# x y z\n# uncertainty\n\n"
                    "10\n1.1191539191370652, 0.523044137695957, 4.781684421973866\n"
                    "1.318541059042735, 1.7085071611126854, 1.4905500967390102\n\n"
                    "8\n-1.5711137587719017, 0.050712492645848636, 5.602895975893089\n"
                    "1.1228681750268494, 1.6589399303511667, 1.5492904185599448\n\n")
    b = _StubFilter()
    b.load_map(str(out))
    assert list(b.landmarks.items()) == [(10, 0), (8, 1)]
    assert np.array_equal(np.asarray(b.pos), np.asarray(a.pos))
    assert np.array_equal(np.asarray(b.unc), np.asarray(a.unc))


def test_golden_map_fixture_parses_with_load_map(golden_dir):
    b = _StubFilter()
    b.load_map(str(golden_dir / "g3_map.txt"))
    g = load_npz("g3_free_run.npz")
    assert [k for k, _ in b.landmarks.items()] == list(g["lm_ids"])
    assert np.allclose(np.asarray(b.pos).ravel(), g["final_state"][10:], rtol=0, atol=0)


def test_process_detections_skips_filter_on_empty_frames():
    calls = []
    from aruco_slam_amd.filters.base_filter import BaseFilter

    class Spy(BaseFilter):
        def observe(self, ids, poses):
            calls.append(list(ids))

        def get_poses(self):
            return np.zeros(10), np.zeros((0, 3))

    s = Spy(np.zeros(10), None)
    s.process_detections(None, np.array([]))
    s.process_detections(np.array([3, 4]), np.zeros((2, 6)))
    assert calls == [[3, 4]]            # base_filter.py:197-204: no predict without detections
    with pytest.raises(NotImplementedError):
        BaseFilter(np.zeros(10), None).get_poses()


def test_small_sequence_shapes():
    from aruco_slam_amd.synthetic import small_sequence
    seq = small_sequence(200, 10, 6, seed=6)
    det = load_npz("c1_detections.npz")
    assert len(seq) == 200 and len(det["timestamps_ms"]) == 200
    n_empty = sum(1 for _, ids, _ in seq if ids is None)
    assert n_empty == 4 and int((~det["has_detections"]).sum()) == 4
    assert max(len(ids) for _, ids, _ in seq if ids is not None) <= 6
    assert any(len(set(ids)) < len(ids) for _, ids, _ in seq if ids is not None)  # duplicates
    cat = np.concatenate([ids for _, ids, _ in seq if ids is not None])
    assert np.array_equal(cat, det["ids"])


def test_trajectory_writer_matches_reference_sample_format(tmp_path):
    """Line format of outputs/trajectory_writer.py:29-40, checked on the first two
    lines of the reference's sample output (/root/reference/outputs/trajectory.txt:1-2)."""
    from aruco_slam_amd.outputs import TrajectoryWriter
    out = tmp_path / "trajectory.txt"
    with TrajectoryWriter(str(out)) as w:
        w.write(0.0, np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]))
        w.write(33.3, np.array([-0.007310123847878528, -0.011780491224478586, 0.02454405259553704,
                                0.9999996317250551, -0.0005479847401364248, 0.0006050155467556792,
                                -0.0002649880505960463, 0, 0, 0]))
        w.write(66.7, np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0]))     # int64 initial pose (D7)
    lines = out.read_text().splitlines()
    assert lines[0] == "0.0000 0.0 0.0 0.0 1.0 0.0 0.0 0.0"
    assert lines[1] == ("0.0333 -0.007310123847878528 -0.011780491224478586 0.02454405259553704 "
                        "0.9999996317250551 -0.0005479847401364248 0.0006050155467556792 "
                        "-0.0002649880505960463")
    assert lines[2] == "0.0667 0 0 0 1 0 0 0"


def test_run_slam_cli_contract():
    from aruco_slam_amd.main import run_slam
    args = run_slam.build_parser().parse_args([])
    assert (args.video, args.filter) == ("input_video.mp4", "ekf")   # run_slam.py:155-168
    with pytest.raises(ValueError, match="Unknown filter type"):
        run_slam.init_tracker("nope", np.zeros(10))
    with pytest.raises(NotImplementedError):
        run_slam.init_tracker("factorgraph", np.zeros(10))
    frames = list(run_slam.detection_frames(str(REPO / "tests" / "golden" / "c1_detections.npz")))
    assert len(frames) == 200 and frames[0][1] is None and frames[1][1] is not None


def test_euler_xyz_to_quat_matches_pinned_oracle():
    """Host-side rvec -> quaternion of EKF_Rotations (ekf_with_rotations.py:216-219) against the
    oracle's restatement, which G5 pins to SciPy through the reference."""
    from aruco_slam_amd.filters.ekf_with_rotations import euler_xyz_to_quat
    from oracle.ekf_numpy import quat_from_euler_xyz
    rng = np.random.default_rng(3)
    ang = rng.uniform(-np.pi, np.pi, size=(64, 3))
    got = euler_xyz_to_quat(ang)
    want = np.stack([quat_from_euler_xyz(a) for a in ang])
    assert np.abs(got - want).max() <= 1e-15
    assert np.abs(np.linalg.norm(got, axis=1) - 1.0).max() <= 1e-15


def test_bench_launches_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` as the driver calls it (no outer launcher, no WORLD_SIZE): the parent becomes the launcher --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` as a
    child process -- relays rank 0's JSON line and returns the child's exit code, before anything touches the GPU."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", REPO / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-7:] == [str(REPO / "bench.py"), "--gpus", "4", "--steps", "20", "--warmup", "5"]
    seen = {}

    def fake_run(argv, env=None, stdout=None, text=None):
        seen["argv"], seen["env"] = argv, env
        return subprocess.CompletedProcess(argv, 3, stdout='noise\n{"metric": "EKF updates/sec", "value": 1.0, "n_gpus": 2}\n')

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "7"])
    with pytest.raises(SystemExit) as stop:
        bench.main()
    assert stop.value.code == 3                                     # the child's exit code
    assert seen["argv"][-4:] == ["--gpus", "2", "--steps", "7"] and "--nproc-per-node=2" in seen["argv"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert capsys.readouterr().out.strip() == '{"metric": "EKF updates/sec", "value": 1.0, "n_gpus": 2}'
    assert bench.parse_args(["--filter", "ekf_rotations"]).landmarks == 400


def test_dictionary_sizes_and_default():
    from aruco_slam_amd.filters.base_filter import dictionary_size
    assert dictionary_size(None) == 50 and dictionary_size(4) == 50        # DICT_5X5_50, base_filter.py:81-82
    assert dictionary_size(7) == 1000 and dictionary_size(16) == 1024 and dictionary_size("x") == 50


def test_hand_counted_kernels_do_not_spill(tmp_path):
    """ekf_cov_update_mfma_f32 issues its loads as inline asm and counts them for s_waitcnt by hand: a
    register spill (scratch traffic on the same counter) would silently break the counting."""
    import re
    import subprocess
    from aruco_slam_amd import _build
    src = _build.CSRC / "ekf_cov_update.hip"
    out = tmp_path / "cov.s"
    subprocess.run([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    str(src), "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S*ekf_cov_update_mfma_f32\S*)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(kernels) >= 12
    assert all(int(spills) == 0 for _, spills in kernels), kernels
    scratch = re.findall(r"\.name:\s+(\S*ekf_cov_update_mfma_f32\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)
    assert all(int(b) == 0 for _, b in scratch), scratch
    # the macro-tile kernel counts its LDS-DMAs and P loads by hand too (scratch accesses would sit in the same counter)
    out2 = tmp_path / "macro.s"
    subprocess.run([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    str(_build.CSRC / "ekf_cov_macro.hip"), "-o", str(out2)], check=True, capture_output=True)
    text2 = out2.read_text()
    macro = re.findall(r"\.name:\s+(\S*ekf_cov_update_macro_f32\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text2)
    assert len(macro) >= 24 and all(int(b) == 0 for _, b in macro), macro


def test_hand_issued_loads_are_not_touched_before_their_wait(tmp_path):
    """The front kernel (16-byte coherent exchange loads) and the f32 covariance update (operand ring)
    issue loads from inline asm; the compiler does not know that their destination registers are in
    flight.  tools/asm_load_hazards.py walks the generated code: no instruction may touch such a register
    between the load and the s_waitcnt that covers it (a copy or spill there would read stale register
    contents whenever memory is slow -- a timing-dependent wrong result)."""
    import subprocess
    import sys
    from concurrent.futures import ThreadPoolExecutor
    from aruco_slam_amd import _build
    sys.path.insert(0, str(_build.PKG.parent / "tools"))
    import asm_load_hazards
    srcs = ["ekf_front.hip", "ekf_front_f64.hip", "ekf_cov_update.hip"]

    def compile_one(name):
        out = tmp_path / (name + ".s")
        subprocess.run([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        str(_build.CSRC / name), "-o", str(out)], check=True, capture_output=True)
        return out

    with ThreadPoolExecutor(max_workers=3) as pool:
        outs = list(pool.map(compile_one, srcs))
    for out in outs:
        n_loads, hazards = asm_load_hazards.scan(str(out))
        if "cov_update" in out.name:
            assert n_loads >= 1000, (out.name, n_loads)      # the f32 covariance update's operand ring
        assert not hazards, (out.name, hazards[:3])


def test_ippe_oracle_recovers_the_poses_its_corners_were_projected_from():
    """oracle/ippe_numpy.py (restatement of cv2.solvePnP(..., SOLVEPNP_IPPE_SQUARE), base_filter.py:92-171; parity
    unpinned: OpenCV is not available and the reference holds no corner / pose pairs): markers projected through
    the reference's calibrated camera model come back with their pose.  The bound is the residual of the 5-iteration
    undistortion at the image border (measured 1.1e-5), not of the pose algebra (1e-12 without distortion)."""
    from scipy.spatial.transform import Rotation
    from conftest import synthetic_marker_views
    from oracle.ippe_numpy import estimate_pose_of_markers, ippe_square, object_points, project_points
    k, dist, corners, tvecs, rots = synthetic_marker_views(300, seed=1)
    poses = estimate_pose_of_markers(corners, 0.16, k, dist)
    for j in range(len(corners)):
        assert np.abs(poses[j, :3] - tvecs[j]).max() <= 5e-5 * np.linalg.norm(tvecs[j])
        assert np.abs((Rotation.from_rotvec(poses[j, 3:]) * rots[j].inv()).as_rotvec()).max() <= 5e-5
    # without distortion the pose algebra is exact to rounding, and the second candidate has the larger error
    px = project_points(object_points(0.16) @ rots[0].as_matrix().T + tvecs[0], k, None)
    t, r, sols = ippe_square(px, 0.16, k, None)
    assert np.abs(t - tvecs[0]).max() <= 1e-10 and sols[0][2] <= sols[1][2]
    assert np.abs((Rotation.from_rotvec(r) * rots[0].inv()).as_rotvec()).max() <= 1e-9
