"""quaff_amd — MI355X-native k-mer-seeded banded pair-HMM DP (the quaff hot path).

The product is the C-ABI shared library quaff_amd/libquaffhip.so (include/quaff_hip.h); this
package is a thin ctypes binding over it plus the build recipe.  There is no CPU fallback:
importing works anywhere, but creating a Context without the library or without a GPU raises.
"""
from .api import Context, DPConfig, QuaffHipError, build_library, library_path, load_library  # noqa: F401
