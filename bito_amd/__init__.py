"""bito_amd -- MI355X-native likelihood engine behind bito's Engine/FatBeagle seam.

The package holds only what the per-tree likelihood path needs: the HIP kernels and
C ABI (csrc/, libbito_amd.so), a ctypes binding, and a host-side mirror of the
reference's Engine / instance interface for that path.
"""
from .engine import BitoAmdError, Engine, PhyloGradient, PhyloModelSpecification, version  # noqa: F401
from .instance import rooted_instance, unrooted_instance  # noqa: F401
from .site_pattern import SitePattern  # noqa: F401
