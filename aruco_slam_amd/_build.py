"""Build the HIP library in-tree (gfx950 only).  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB_PATH = LIB_DIR / "libekf_slam_hip.so"
SOURCES = ["ekf_api.hip", "ekf_small_kernels.hip", "ekf_front.hip", "ekf_front_f64.hip", "ekf_cov_update.hip", "ekf_cov_macro.hip", "ekf_pose_ippe.hip"]
HEADERS = ["ekf_device.h", "ekf_kernels.h", "ekf_solve_device.h", "ekf_solve_big.h", "ekf_front_impl.h", "../../include/ekf_slam_hip.h"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need /opt/rocm)")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = PKG / "build"
    obj_dir.mkdir(exist_ok=True)
    hdrs = [CSRC / h for h in HEADERS]
    objs = []
    flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
    jobs = []
    for src in SOURCES:
        obj = obj_dir / (Path(src).stem + ".o")
        if force or _stale(obj, [CSRC / src, *hdrs]):
            jobs.append([hipcc(), *flags, "-c", str(CSRC / src), "-o", str(obj)])
        objs.append(obj)
    if jobs:        # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as pool:
            list(pool.map(run, jobs))
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH),
               *map(str, objs)]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True))
