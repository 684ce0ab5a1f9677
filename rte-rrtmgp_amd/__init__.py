"""MI355X-native RTE+RRTMGP compute path (hand-written HIP for gfx950 behind the reference's
bind(C) kernel interface).  See DESIGN.md / INTEGRATION.md.

Sub-modules
  cabi      ctypes binding table of include/rte_rrtmgp_kernels.h (shared by every library that
            exports that ABI)
  hiplib    loader/builder of librte_rrtmgp_hip.so -- the product; fails loudly if it is missing
  frontend  host-side mirror of the reference frontend's calls on this path
            (ty_gas_optics_rrtmgp%gas_optics, rte_lw, rte_sw) over the C ABI
  synth     seeded synthetic k-distribution / atmosphere generators
"""
from . import cabi, synth  # noqa: F401
from . import frontend, hiplib, sharding  # noqa: F401

__all__ = ["cabi", "synth", "frontend", "hiplib", "sharding"]
