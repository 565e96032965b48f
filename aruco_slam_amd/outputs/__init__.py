from .trajectory_writer import TrajectoryWriter  # noqa: F401
