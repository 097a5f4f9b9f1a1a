"""phylomap_amd: MI355X-native stochastic-mapping engine behind phylomap's sumstat* API.

Host side mirrors the reference's R-level interface (R/sumstat*.R); the compute path is the HIP
library built from phylomap_amd/csrc (C-ABI in include/phylomap_hip.h).  There is no CPU fallback:
every sumstat* call fails loudly if the HIP library is missing.
"""
from .treeorder import pruningwiseedgeorder, makenodelist, myreorder  # noqa: F401

__all__ = ["pruningwiseedgeorder", "makenodelist", "myreorder"]
