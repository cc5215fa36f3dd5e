"""magnetite_amd -- MI355X-native solver hot path of kyle-tennison/Magnetite.

csrc/      hand-written HIP kernels (gfx950) + the C ABI of include/magnetite_hip.h
solver.py  host-side mirror of the reference's solver interface (solver.rs:543-547)
meshgen.py synthetic meshes / boundary-rule stamping for the BASELINE.json configs
"""
from . import _lib, meshgen, solver  # noqa: F401
from ._lib import build  # noqa: F401
from .solver import (Context, Element, MagnetiteError, ModelMetadata, Node, Vertex,  # noqa: F401
                     compute_element_area, run)

__all__ = ["Context", "Element", "MagnetiteError", "ModelMetadata", "Node", "Vertex", "compute_element_area",
           "run", "build", "meshgen", "solver"]
