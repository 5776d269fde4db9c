"""kmtricks_amd -- MI355X-native counting/merge engine for the kmtricks hot path.

The product is libkmx.so (hand-written HIP for gfx950 behind the C ABI of
include/kmx.h).  This package is only the Python binding used by tests/ and
bench.py; it never computes anything itself and has no CPU fallback: importing
`kmtricks_amd.lib` raises if libkmx.so has not been built, and creating a
context raises if there is no HIP device.
"""
__version__ = "0.1.0"
