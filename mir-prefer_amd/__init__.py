"""mir-prefer_amd: MI355X-native candidate -> fold -> predict hot path of miR-PREFeR.

Host side mirrors the reference's own stage functions (miR_PREFeR.py) and calls hand-written
HIP kernels through the ctypes C-ABI of libmirprefer.so (include/mirprefer.h)."""
__version__ = "0.1.0"
