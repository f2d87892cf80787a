"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes binding of the CPU oracle (oracle/liboracle.so).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
nothing under tetrex_amd/ does.
"""
from .binding import *  # noqa: F401,F403
