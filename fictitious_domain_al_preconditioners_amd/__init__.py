"""MI355X-native augmented-Lagrangian FGMRES hot path.

Sub-modules: ``solver`` (ctypes front-end of the C ABI in include/alfd/alfd.h),
``problems`` (synthetic fictitious-domain operators), ``prm`` (deal.II .prm
subset reader), ``_abi`` (struct mirrors).
"""
import os

# The host-side OpenMP helpers (generator; the test oracle) must not spawn one
# thread per hardware thread inside a CPU-limited container: cap unless told.
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, os.cpu_count() or 1))))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

__version__ = "0.1.0"
