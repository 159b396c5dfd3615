"""mocapv2_amd -- MI355X (gfx950) implementation of MocapV2's per-frame hot path.

IR blob detection / centroid extraction and multi-view epipolar correspondence + DLT triangulation, as
hand-written HIP kernels behind a C-ABI (include/mocap_hip.h, mocapv2_amd/libmocap_hip.so), with the
reference's Python call surface mirrored in mocapv2_amd.lib (ImageOperations, CudaOperations, Helpers).

There is no CPU fallback: everything that computes needs the HIP library and a GPU and raises otherwise.
"""
__version__ = "0.1.0"
