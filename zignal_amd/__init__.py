"""zignal_amd -- MI355X (gfx950) evaluator for Flowz signal flow-graphs.

Host-side mirror of the reference's EDSL (`zignal_amd.flowz`) over the C ABI of
`zignal_amd/lib/libflowz_hip.so` (include/flowz_hip.h).  Importing the package loads the HIP
library and fails loudly if it is missing: there is no CPU or PyTorch fallback.
"""
from . import flowz  # noqa: F401
from ._capi import FlowzError, NoDeviceError, LIB_PATH  # noqa: F401

__all__ = ["flowz", "FlowzError", "NoDeviceError", "LIB_PATH"]
