"""ctypes binding of the C ABI in ``include/ekf_slam_hip.h``.

PyTorch is used for exactly three things: owning the device buffers
(covariance, state, workspace, resident detection arrays), providing the HIP
stream, and ``torch.distributed`` for the final gather.  All arithmetic runs in
the HIP library; there is no CPU fallback -- a missing library or a missing GPU
raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from ._build import LIB_PATH

EKF_COV_F64, EKF_COV_F32 = 0, 1
EKF_QUAT_AS_WRITTEN, EKF_QUAT_SCALAR_FIRST = 0, 1
EKF_COVK_AUTO, EKF_COVK_VALU, EKF_COVK_MFMA, EKF_COVK_MFMA_TILE, EKF_COVK_MFMA_MACRO = 0, 1, 2, 3, 4

# every symbol include/ekf_slam_hip.h declares
EXPORTED_SYMBOLS = (
    "ekf_default_config", "ekf_query_sizes", "ekf_create", "ekf_destroy", "ekf_bind_buffers",
    "ekf_reset", "ekf_grow", "ekf_add_markers", "ekf_observe", "ekf_observe_device",
    "ekf_observe_sequence_device", "ekf_last_sequence_mode", "ekf_get_camera", "ekf_get_state", "ekf_get_cov_diag",
    "ekf_get_cov", "ekf_set_state", "ekf_set_cov", "ekf_num_landmarks", "ekf_sync",
    "ekf_set_fused", "ekf_set_kernel_timing", "ekf_get_kernel_timing", "ekf_debug_fetch",
    "ekf_estimate_poses_device", "ekf_estimate_poses", "ekf_last_error_string",
)


class EkfConfig(C.Structure):
    _fields_ = [
        ("max_landmarks", C.c_int32), ("max_visible", C.c_int32), ("cov_dtype", C.c_int32),
        ("quat_mode", C.c_int32), ("cov_kernel", C.c_int32), ("model", C.c_int32), ("reserved", C.c_int32), ("flags", C.c_int32),
        ("initial_camera_uncertainty", C.c_double), ("initial_landmark_uncertainty", C.c_double),
        ("r_uncertainty", C.c_double), ("q_cam", C.c_double), ("q_err", C.c_double),
        ("q_lm", C.c_double), ("stream", C.c_void_p),
    ]


class EkfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ekf_slam_hip error {code}: {msg}")
        self.code = code


_lib = None


def load_library(path: str | Path | None = None):
    """dlopen the in-tree library; raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise FileNotFoundError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the EKF update path)")
    # torch ships its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Import
    # torch FIRST so the library below binds to that already-loaded runtime: a
    # second HIP runtime in the process sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(str(p))
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    sig = {
        "ekf_default_config": [C.POINTER(EkfConfig)],
        "ekf_query_sizes": [C.POINTER(EkfConfig), C.POINTER(C.c_int64), C.POINTER(C.c_size_t),
                            C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
        "ekf_create": [C.POINTER(EkfConfig), C.POINTER(vp)],
        "ekf_destroy": [vp],
        "ekf_bind_buffers": [vp, vp, C.c_int64, vp, vp, C.c_size_t],
        "ekf_reset": [vp, dp],
        "ekf_grow": [vp, C.c_int32, C.c_int32, vp, C.c_int64, vp, vp, C.c_size_t],
        "ekf_add_markers": [vp, dp, dp, C.c_int32],
        "ekf_observe": [vp, ip, dp, C.c_int32],
        "ekf_observe_device": [vp, vp, vp, C.c_int32],
        "ekf_observe_sequence_device": [vp, vp, vp, C.c_int32, C.c_int32, vp],
        "ekf_get_camera": [vp, dp],
        "ekf_get_state": [vp, dp, C.c_int32],
        "ekf_get_cov_diag": [vp, dp, C.c_int32],
        "ekf_get_cov": [vp, dp, C.c_int32],
        "ekf_set_state": [vp, dp, C.c_int32],
        "ekf_set_cov": [vp, dp, C.c_int32],
        "ekf_num_landmarks": [vp],
        "ekf_last_sequence_mode": [vp],
        "ekf_sync": [vp],
        "ekf_set_fused": [vp, C.c_int32],
        "ekf_set_kernel_timing": [vp, C.c_int32],
        "ekf_get_kernel_timing": [vp, C.c_int32, dp, C.POINTER(C.c_int64)],
        "ekf_debug_fetch": [vp, C.c_int32, dp, C.c_size_t],
        "ekf_estimate_poses_device": [vp, C.c_int32, C.c_double, dp, dp, C.c_int32, vp, vp],
        "ekf_estimate_poses": [dp, C.c_int32, C.c_double, dp, dp, C.c_int32, dp, vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.ekf_last_error_string.argtypes = []
    lib.ekf_last_error_string.restype = C.c_char_p
    if path is None:
        _lib = lib
    return lib


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def estimate_poses(corners, marker_size: float, camera_matrix, dist_coeffs=None, device="cuda:0") -> np.ndarray:
    """Batched IPPE-square pose of every detected marker on the GPU (ekf_estimate_poses; replaces the per-marker
    cv2.solvePnP loop of base_filter.py:92-171).  corners: array-like [m,4,2] (or cv2's list of [1,4,2]) of pixel
    coordinates; returns [m,6] = [tvec | rvec].  No CPU fallback."""
    import torch
    lib = load_library()
    if not torch.cuda.is_available():
        raise RuntimeError("the pose front end needs a HIP device (no CPU fallback)")
    c = np.ascontiguousarray(np.asarray(corners, dtype=np.float64).reshape(-1, 4, 2))
    k = np.ascontiguousarray(np.asarray(camera_matrix, dtype=np.float64).reshape(3, 3))
    d = np.ascontiguousarray(np.asarray([] if dist_coeffs is None else dist_coeffs, dtype=np.float64).reshape(-1))
    out = np.zeros((c.shape[0], 6))
    with torch.cuda.device(torch.device(device)):
        rc = lib.ekf_estimate_poses(_dptr(c), c.shape[0], float(marker_size), _dptr(k), _dptr(d) if d.size else None,
                                    int(d.size), _dptr(out), None)
    if rc != 0:
        raise EkfError(rc, lib.ekf_last_error_string().decode())
    return out


class HipEkf:
    """One filter instance = one C handle + the torch tensors it borrows."""

    KERNEL_NAMES = ("gather", "solve", "panel", "cov_update")
    MAX_VISIBLE_LIMIT = {3: 64, 10: 50}      # detections per frame the kernels take (by landmark width: EKF / EKF_Rotations)

    def __init__(self, max_landmarks: int, max_visible: int, cov_dtype="float64",
                 quat_mode="as_written", cov_kernel="auto", device="cuda:0", noise=None,
                 lookahead=None, model="ekf", fused=True):
        import torch
        self._torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("the EKF update path needs a HIP device (no CPU fallback)")
        self.device = torch.device(device)
        cfg = EkfConfig()
        self._check(self.lib.ekf_default_config(C.byref(cfg)))
        cfg.max_landmarks = int(max_landmarks)
        cfg.max_visible = int(max_visible)
        cfg.cov_dtype = {"float64": EKF_COV_F64, "float32": EKF_COV_F32}[str(cov_dtype)]
        cfg.quat_mode = {"as_written": EKF_QUAT_AS_WRITTEN,
                         "scalar_first": EKF_QUAT_SCALAR_FIRST}[quat_mode]
        kern = {"auto": EKF_COVK_AUTO, "valu": EKF_COVK_VALU, "mfma": EKF_COVK_MFMA, "mfma_tile": EKF_COVK_MFMA_TILE,
                "mfma_macro": EKF_COVK_MFMA_MACRO}
        cfg.cov_kernel = kern[cov_kernel]
        cfg.flags = {None: 0, False: 1, True: 2}[lookahead]   # None: pipelined sequence mode where it wins (by size); False: never; True: always
        if not fused:
            cfg.flags |= 4        # separate gather / solve / panel launches
        self.fused = bool(fused)  # ("force" of earlier versions == True: there is no automatic fallback any more)
        self._last_m = 1
        cfg.model = {"ekf": 0, "ekf_rotations": 1}[model]
        self.lm_dims, self.rows_per_detection = (10, 7) if cfg.model == 1 else (3, 3)
        for key, val in (noise or {}).items():
            setattr(cfg, key, float(val))
        self.cov_dtype = str(cov_dtype)
        self.max_landmarks, self.max_visible = int(max_landmarks), int(max_visible)
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream(device=self.device)
            cfg.stream = self.stream.cuda_stream
            ld, cb, sb, wb = C.c_int64(), C.c_size_t(), C.c_size_t(), C.c_size_t()
            self._check(self.lib.ekf_query_sizes(C.byref(cfg), C.byref(ld), C.byref(cb),
                                                 C.byref(sb), C.byref(wb)))
            self.ld = ld.value
            tdt = torch.float64 if cfg.cov_dtype == EKF_COV_F64 else torch.float32
            self.cov_t = torch.zeros((self.ld, self.ld), dtype=tdt, device=self.device)
            self.state_t = torch.zeros((sb.value // 8,), dtype=torch.float64, device=self.device)
            self.ws_t = torch.zeros((wb.value,), dtype=torch.uint8, device=self.device)
            torch.cuda.synchronize(self.device)
            handle = C.c_void_p()
            self._check(self.lib.ekf_create(C.byref(cfg), C.byref(handle)))
            self.h = handle
            self._check(self.lib.ekf_bind_buffers(self.h, self.cov_t.data_ptr(), self.ld,
                                                  self.state_t.data_ptr(), self.ws_t.data_ptr(),
                                                  wb.value))
        self.cfg = cfg

    def _check(self, rc):
        if rc != 0:
            raise EkfError(rc, self.lib.ekf_last_error_string().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.ekf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- filter operations -------------------------------------------------
    @property
    def num_landmarks(self) -> int:
        return self.lib.ekf_num_landmarks(self.h)

    @property
    def dims(self) -> int:
        return self.lm_dims * self.num_landmarks + 10

    def reset(self, initial_pose):
        p = np.ascontiguousarray(initial_pose, dtype=np.float64)
        assert p.shape == (10,)
        self._check(self.lib.ekf_reset(self.h, _dptr(p)))

    def grow(self, new_max_landmarks: int | None = None, new_max_visible: int | None = None):
        """Move the filter into buffers for a larger capacity (ekf_grow): new tensors sized by ekf_query_sizes, the library
        copies state / covariance device to device and re-binds; the old tensors are released afterwards."""
        torch = self._torch
        cfg = self.cfg
        new_max_landmarks = self.max_landmarks if new_max_landmarks is None else int(new_max_landmarks)
        new_max_visible = self.max_visible if new_max_visible is None else int(new_max_visible)
        ncfg = EkfConfig()
        C.memmove(C.byref(ncfg), C.byref(cfg), C.sizeof(EkfConfig))
        ncfg.max_landmarks = int(new_max_landmarks)
        ncfg.max_visible = int(new_max_visible)
        with torch.cuda.device(self.device):
            ld, cb, sb, wb = C.c_int64(), C.c_size_t(), C.c_size_t(), C.c_size_t()
            self._check(self.lib.ekf_query_sizes(C.byref(ncfg), C.byref(ld), C.byref(cb), C.byref(sb), C.byref(wb)))
            cov_t = torch.empty((ld.value, ld.value), dtype=self.cov_t.dtype, device=self.device)
            state_t = torch.empty((sb.value // 8,), dtype=torch.float64, device=self.device)
            ws_t = torch.empty((wb.value,), dtype=torch.uint8, device=self.device)
            torch.cuda.synchronize(self.device)
            self._check(self.lib.ekf_grow(self.h, int(new_max_landmarks), int(new_max_visible), cov_t.data_ptr(), ld.value,
                                          state_t.data_ptr(), ws_t.data_ptr(), wb.value))
        self.cov_t, self.state_t, self.ws_t, self.ld = cov_t, state_t, ws_t, ld.value
        self.cfg = ncfg
        self.max_landmarks, self.max_visible = int(new_max_landmarks), int(new_max_visible)

    def add_markers(self, xyz, uncertainty=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 6 if self.lm_dims == 10 else 3)
        if self.num_landmarks + xyz.shape[0] > self.max_landmarks:      # the reference appends without limit (:274-290)
            self.grow(max(2 * self.max_landmarks, self.num_landmarks + xyz.shape[0]))
        unc = None
        if uncertainty is not None:
            unc = np.ascontiguousarray(
                np.broadcast_to(np.asarray(uncertainty, dtype=np.float64), (xyz.shape[0], self.lm_dims)))
        self._check(self.lib.ekf_add_markers(self.h, _dptr(xyz), _dptr(unc) if unc is not None else None,
                                             xyz.shape[0]))

    def observe(self, lm_index, z):
        idx = np.ascontiguousarray(lm_index, dtype=np.int32)
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1, self.rows_per_detection)
        assert idx.shape[0] == z.shape[0]
        self._last_m = idx.shape[0]
        if idx.shape[0] > self.max_visible:      # the reference takes any number of detections per frame (:158-200)
            self.grow(new_max_visible=min(self.MAX_VISIBLE_LIMIT[self.lm_dims], max(2 * self.max_visible, idx.shape[0])))
        self._check(self.lib.ekf_observe(self.h, idx.ctypes.data_as(C.POINTER(C.c_int32)), _dptr(z),
                                         idx.shape[0]))

    def observe_sequence(self, idx_t, z_t, traj_t=None):
        """idx_t int32 [F,m], z_t float64 [F,m,3] device tensors (resident
        detections); traj_t float64 [F,7] or None."""
        frames, m = idx_t.shape
        assert idx_t.is_cuda and z_t.is_cuda and idx_t.is_contiguous() and z_t.is_contiguous()
        assert tuple(z_t.shape) == (frames, m, self.rows_per_detection)
        self._last_m = m
        self._check(self.lib.ekf_observe_sequence_device(
            self.h, idx_t.data_ptr(), z_t.data_ptr(), m, frames,
            traj_t.data_ptr() if traj_t is not None else None))

    SEQUENCE_MODES = {0: "none", 1: "serial", 2: "pipelined", 3: "serial (the two streams share one hardware queue)",
                      4: "serial (another handle of the process is pipelining)"}

    def last_sequence_mode(self) -> str:
        """What the last observe_sequence call did (ekf_last_sequence_mode)."""
        return self.SEQUENCE_MODES[self.lib.ekf_last_sequence_mode(self.h)]

    def sync(self):
        self._check(self.lib.ekf_sync(self.h))

    def get_state(self, count=None):
        n = self.dims if count is None else count
        out = np.empty(n)
        self._check(self.lib.ekf_get_state(self.h, _dptr(out), n))
        return out

    def get_cov_diag(self):
        out = np.empty(self.dims)
        self._check(self.lib.ekf_get_cov_diag(self.h, _dptr(out), self.dims))
        return out

    def get_cov(self):
        n = self.dims
        out = np.empty((n, n))
        self._check(self.lib.ekf_get_cov(self.h, _dptr(out), n))
        return out

    def set_state_cov(self, state, cov):
        state = np.ascontiguousarray(state, dtype=np.float64)
        n_lm = (state.shape[0] - 10) // self.lm_dims
        if n_lm > self.max_landmarks:
            self.grow(n_lm)
        self._check(self.lib.ekf_set_state(self.h, _dptr(state), n_lm))
        cov = np.ascontiguousarray(cov, dtype=np.float64)
        assert cov.shape == (state.shape[0], state.shape[0])
        self._check(self.lib.ekf_set_cov(self.h, _dptr(cov), state.shape[0]))

    # -- instrumentation -----------------------------------------------------
    def set_kernel_timing(self, enable):
        """False/0 off, True/1 all kernels, 2 covariance update only."""
        self._check(self.lib.ekf_set_kernel_timing(self.h, int(enable)))

    def kernel_timing(self):
        out = {}
        for i, name in enumerate(self.KERNEL_NAMES):
            us, cnt = C.c_double(), C.c_int64()
            self._check(self.lib.ekf_get_kernel_timing(self.h, i, C.byref(us), C.byref(cnt)))
            out[name] = (us.value, cnt.value)
        if self.fused:
            # one launch: slot 0 is the whole front kernel, slots 1-2 are empty event gaps
            out = {"front": out["gather"], "cov_update": out["cov_update"]}
        return out

    def set_fused(self, enable: bool):
        """Fused front kernel (True) or the three stage kernels (False) from the next frame on."""
        self._check(self.lib.ekf_set_fused(self.h, int(bool(enable))))
        self.fused = bool(enable)

    def debug_enable_w(self):
        dummy = np.zeros(1)
        self._check(self.lib.ekf_debug_fetch(self.h, -1, _dptr(dummy), 1))

    def debug_enable_stamps(self, light: bool = False):
        """In-kernel time stamps of the front kernel without the debug copies of W / L (production code path).
        `light`: only the stamps at the start / end of the roles (the others perturb what they measure)."""
        dummy = np.zeros(1)
        self._check(self.lib.ekf_debug_fetch(self.h, -3 if light else -2, _dptr(dummy), 1))

    def debug_fetch(self, what: str, m: int):
        rd = self.rows_per_detection
        k, kp, n = rd * m, -(-rd * m // 16) * 16, self.dims
        shape = {"jac": (k, 20 if rd == 7 else 13), "resid": (k,), "L": (kp, kp), "W": (kp, n), "A": (k, n),
                 "stamps": (64,), "cov_stats": (32,)}[what]
        code = {"jac": 0, "resid": 1, "L": 2, "W": 3, "A": 4, "stamps": 5, "cov_stats": 6}[what]
        out = np.empty(shape)
        self._check(self.lib.ekf_debug_fetch(self.h, code, _dptr(out), out.size))
        return out
