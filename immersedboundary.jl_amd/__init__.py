"""ibamd -- MI355X-native residual hot path for ImmersedBoundary.jl-style block-octree domains.

Host-side mirror of the reference's ``Domain``/``Partition`` operator API
(/root/reference/src/ImmersedBoundary.jl) for the per-partition residual sweep; all compute is
in libibhip.so (hand-written HIP for gfx950) behind the C ABI of include/ibhip.h.

Importing this package does not need a GPU (mesh/domain construction is host-side); anything
that computes needs libibhip.so and a gfx950 device and fails loudly otherwise.
"""
from .mesher import (Ball, Box, DistanceField, Line, Mesh, Stereolitography, cat, centers_and_normals,
                     feature_regions, get_cells, merge_points, refine_to_length)
from .accumulator import Accumulator
from .domain import Boundary, Domain, Partition, Surface, multigrid


def __getattr__(name):
    # GPU-facing names are resolved lazily so that `import ibamd` works where torch is absent
    import importlib
    if name.startswith("__"):
        raise AttributeError(name)
    backend = importlib.import_module(__name__ + ".backend")
    if hasattr(backend, name):
        return getattr(backend, name)
    if name == "HipArray":
        return getattr(importlib.import_module(__name__ + ".hiparray"), name)
    if name == "FAS":
        return importlib.import_module(__name__ + ".solver").FAS
    raise AttributeError(name)
