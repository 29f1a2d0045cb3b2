"""idocp_amd -- MI355X-native KKT-condensation + Riccati hot path of idocp.

The product is the C-ABI shared library idocp_amd/lib/libidocp_hip.so
(include/idocp_hip.h) and the C++ facade in include/idocp/.  This package only
carries the build helper and the ctypes binding used by tests and bench.py.
"""
from . import capi  # noqa: F401
