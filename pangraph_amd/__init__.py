"""pangraph_amd -- MI355X-native pairwise block-alignment backend.

Importing the package only sets HIP runtime defaults; the entry points live in batch.py (native batch C-ABI),
mm2ffi.py (the minimap2-sys shaped C-ABI) and dist.py (multi-GPU match-list gather).
"""
import os

# The extension stage keeps several independent persistent launches in flight (DP classes x concurrent query sets, each on
# its own stream).  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and launches that share a
# queue run back to back: 6 queues measured 6 % faster per step than 4 on MI355X, 12 and more slow every kernel down.
# Only effective when set before the process initialises HIP, hence here; a host that loads libpgalign.so directly exports
# the variable itself.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
