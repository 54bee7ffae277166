"""channeld_amd — MI355X-native (HIP / gfx950) spatial interest management and
fan-out for channeld's SpatialChannel hot path.

The package is a thin host mirror of the reference's Go `SpatialController`
interface over the C-ABI of libchd_spatial.so (include/chd_spatial.h).  All
computation happens in hand-written HIP kernels; importing the package without
the built library, or using it without a gfx950 device, raises.
"""
from ._lib import ChdError, LIB_PATH, load  # noqa: F401
from .controller import (  # noqa: F401
    BoxAOI, ConeAOI, SpatialError, SpatialInfo, SpatialInterestQuery, SpatialRegion, SphereAOI, SpotsAOI,
    StaticGrid2DSpatialController,
)
from .engine import SpatialWorld, TickResult, UpdateBatch  # noqa: F401

__all__ = [
    "StaticGrid2DSpatialController", "SpatialWorld", "TickResult", "UpdateBatch", "SpatialInfo", "SpatialInterestQuery",
    "SpotsAOI", "BoxAOI", "SphereAOI", "ConeAOI", "SpatialRegion", "SpatialError", "ChdError", "load",
]
