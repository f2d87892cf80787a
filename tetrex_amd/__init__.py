"""tetrex_amd — MI355X-native TetRex query path.

The product is native: hand-written HIP kernels for gfx950 behind the C-ABI of
``include/txq.h`` (``tetrex_amd/libtxq.so``) and a C++ host front-end
(``tetrex_amd/libtetrex_host.so`` + the ``tetrex`` CLI).  This Python package is only the thin
ctypes view of those libraries used by tests and ``bench.py``; it contains no compute and
no CPU fallback — importing :mod:`tetrex_amd.capi` fails loudly when the HIP library has
not been built.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
