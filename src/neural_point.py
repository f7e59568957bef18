"""`from src.neural_point import NeuralPointCloud`."""
from loopy_slam_amd.slam import NeuralPointCloud  # noqa: F401
