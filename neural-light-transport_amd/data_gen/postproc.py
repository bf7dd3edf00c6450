"""data_gen/postproc.py:53-82 on HIP: UV albedo, diffuse bases, and their camera-space remap."""
from .. import _capi as C
from .util import remap


def compute_albedo(rgb_uv_frames):
    """postproc.py:53-64: rgb_uv_frames [F,H,W,3] uint8 (every trainvali rgb.png) -> float64 [H,W,3]."""
    return C.albedo(rgb_uv_frames.contiguous())


def compute_diffuse_bases(albedo, lvis_frames, uv2cam=None):
    """postproc.py:66-82: diffuse.png = albedo * lvis (uint8, truncating) for every frame; with
    uv2cam [F,imh,imw,2] (fp16 as stored) also diffuse_camspc.png.  Returns (diffuse, diffuse_camspc)."""
    diffuse = C.diffuse_base(albedo, lvis_frames.contiguous())
    cam = None
    if uv2cam is not None:
        cam = [remap(diffuse[f], uv2cam[f]) for f in range(diffuse.shape[0])]
    return diffuse, cam
