"""Trajectory text sink: same file format as the reference's
``TrajectoryWriter`` (/root/reference/outputs/trajectory_writer.py:16-51).

One line per frame: ``f"{ms/1000:.4f} x y z p3 p4 p5 p6"`` where ``p3..p6`` are
``state[3:7]`` = qw qx qy qz (the reference's own comment calls this TUM order;
it is not -- SURVEY appendix A, D8).  Numbers are written with ``str()`` of the
array element, so the integer initial pose prints as ``0``/``1`` until the
first marker has been added (D7).
"""
from __future__ import annotations

from pathlib import Path


class TrajectoryWriter:
    def __init__(self, filename: str) -> None:
        self.file = None
        self.filename = filename

    def __enter__(self):
        self.file = Path(self.filename).open("w", encoding="utf-8")
        return self

    def write(self, timestamp, pose) -> None:
        quat = pose[3:]
        seconds = timestamp / 1000
        if self.file:
            line = f"{seconds:.4f} "
            line += f"{pose[0]} {pose[1]} {pose[2]} "
            line += f"{quat[0]} {quat[1]} {quat[2]} {quat[3]}\n"
            self.file.write(line)

    def __exit__(self, exc_type, exc_value, traceback) -> None:
        if self.file:
            self.file.close()
            self.file = None
