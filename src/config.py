"""`from src import config` (run.py:7): load_config / update_recursive / get_model of loopy_slam_amd.config."""
from loopy_slam_amd.config import load_config, update_recursive, get_model  # noqa: F401
