"""metheor_amd -- MI355X (gfx950) engine for metheor's per-read CpG-pattern hot path.

The product is the C-ABI shared library `libmetheor_hip.so` (include/metheor_hip.h) and the C++
host around it.  This Python package is only the thin ctypes binding that tests and bench.py call
through; it never computes anything itself and has NO CPU fallback: if the HIP library is missing
or no gfx950 device is present, it raises.
"""
from .capi import (Batch, Engine, MthError, PdrLpmdParams, allreduce_lpmd, device_count, lib, library_path)  # noqa: F401

__all__ = ["Batch", "Engine", "MthError", "PdrLpmdParams", "allreduce_lpmd", "device_count", "lib", "library_path"]
