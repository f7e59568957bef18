"""`from src.Mapper import Mapper`."""
from loopy_slam_amd.slam import Mapper  # noqa: F401
