"""Host mirror of the reference's offline buffer assembly (data_gen/*.py) on libnlt_hip.so.
Same function names and argument meaning as the reference; tensors are torch CUDA tensors.
The Blender / Cycles parts (rendering, ray casting, BVH shadow rays, smart-UV unwrap) are out of
scope: their OUTPUTS (per-pixel hit position / normal / face index, occlusion flags, the unwrap
table) are this package's inputs."""
from . import util, render, get_neighbors, postproc      # noqa: F401
