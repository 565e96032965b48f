"""Camera-trajectory text sink.

Interface and byte format are the contract of the reference's ``TrajectoryWriter``
(/root/reference/outputs/trajectory_writer.py:16-51) as ``main/run_slam.py:110-125`` uses it:
a context manager with ``write(timestamp_ms, pose)``; one line per frame,

    <timestamp in seconds, 4 decimals> x y z qw qx qy qz

The four values after the position are ``pose[3:7]`` in the filter's own order, scalar first (the
reference labels the line "TUM"; TUM order would be qx qy qz qw -- SURVEY appendix A, D8).  Every
number except the time stamp is written with ``str()`` of the array element: shortest round-trip
repr for floats, and plain ``0`` / ``1`` for the integer initial pose that the filter hands out until
the first marker has been added (D7).  ``tests/test_host_cpu.py`` pins the format on the reference's
sample output.
"""
from __future__ import annotations

import contextlib
import io


class TrajectoryWriter(contextlib.AbstractContextManager):
    """``with TrajectoryWriter(path) as w: w.write(ms, pose)``.  Outside the ``with`` block (before
    entering, after leaving) ``write`` is a no-op, as in the reference."""

    def __init__(self, filename: str) -> None:
        self.filename = filename
        self._sink: io.TextIOBase | None = None

    def __enter__(self) -> "TrajectoryWriter":
        self._sink = open(self.filename, "w", encoding="utf-8")
        return self

    def write(self, timestamp, pose) -> None:
        if self._sink is None:
            return
        fields = [f"{timestamp / 1000:.4f}", *(str(pose[i]) for i in range(7))]
        self._sink.write(" ".join(fields) + "\n")

    def __exit__(self, *exc_info) -> None:
        sink, self._sink = self._sink, None
        if sink is not None:
            sink.close()
        return None
