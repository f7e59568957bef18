"""`from src.Point_SLAM import Point_SLAM` (run.py:8)."""
from loopy_slam_amd.slam import Point_SLAM  # noqa: F401
