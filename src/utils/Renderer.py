"""`from src.utils.Renderer import Renderer`."""
from loopy_slam_amd.slam import Renderer  # noqa: F401
