"""MI355X-native implementation of the DRAM DC3D forward/backward hot path.

Host side of libdram_hip.so: ctypes binding (`_lib`), autograd Functions
(`functional`) and nn.Module leaves (`modules`).  The drop-in modules that mirror
the reference's flat `parts.py` / `models.py` live one directory up.
"""
from . import _lib, functional, modules  # noqa: F401

__all__ = ["_lib", "functional", "modules"]
