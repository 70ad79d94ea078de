"""MI355X-native volumetric feature aggregation (the un-projection hot path of MultiviewHMR).

Public surface mirrors the reference modules it replaces:

    multiviewhmr_amd.aggregation   <->  models/aggregation.py   (unprojection, VolumeGenerator, build_volume_generator)
    multiviewhmr_amd.multiview     <->  utils/multiview.py      (Camera, projection helpers, DLT triangulation)
    multiviewhmr_amd.volumetric    <->  utils/volumetric.py     (Cuboid3D, get_rotation_matrix, rotate_coord_volume)

The compute lives in lib/libmvhmr_unproject.so (hand-written HIP for gfx950 behind the C ABI of
include/mvhmr_unproject.h); there is no CPU or PyTorch fallback -- importing is cheap, calling
`unprojection` without the built library or without a HIP device raises.
"""
__version__ = "0.1.0"

from . import aggregation, multiview, volumetric  # noqa: F401
from .aggregation import VolumeGenerator, build_volume_generator, unprojection  # noqa: F401
