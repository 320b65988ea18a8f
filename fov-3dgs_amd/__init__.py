"""fovraster: MI355X-native foveated 3D-Gaussian-splatting rasterizer (hot path of Fov-3DGS).

Python/PyTorch-ROCm host code over a C-ABI HIP library (csrc/ -> libfovraster_hip.so).
See DESIGN.md for the path, the boundary and the kernels.
"""
__version__ = "0.1.0"
