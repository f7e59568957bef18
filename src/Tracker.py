"""`from src.Tracker import Tracker`."""
from loopy_slam_amd.slam import Tracker  # noqa: F401
