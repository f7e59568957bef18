"""`from src.utils.Logger import Logger`."""
from loopy_slam_amd.slam import Logger  # noqa: F401
