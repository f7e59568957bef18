"""loopy_slam_amd — MI355X (gfx950) native render/optimise hot path of Loopy-SLAM.

HIP kernels behind a C ABI (include/loopy_hip.h, csrc/), driven from Python on PyTorch-ROCm.
The package never falls back to a CPU implementation: importing is cheap, but any compute
call raises unless libloopyhip.so is built and a HIP device is visible.
"""
__version__ = '0.1.0'
