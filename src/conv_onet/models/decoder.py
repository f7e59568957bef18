"""`from src.conv_onet.models.decoder import NICER`."""
from loopy_slam_amd.slam import NICER  # noqa: F401
