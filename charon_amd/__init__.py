"""charon_amd -- MI355X-native implementation of the per-read classification path of `charon dehost`.

The product is libcharon_hip.so (HIP kernels + C ABI, include/charon_hip.h) and the C++14 host front
end in charon_amd/csrc.  This package is only the thin ctypes binding used by tests and bench.py; it
has no CPU fallback: importing `charon_amd.api` without the built library raises.
"""
__version__ = "0.1.0"
