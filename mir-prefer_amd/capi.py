"""ctypes binding of libmirprefer.so (include/mirprefer.h).

There is no CPU fallback: if the shared library is missing or no GPU is usable, the calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmirprefer.so")


class MirpError(RuntimeError):
    pass


class FoldLine(C.Structure):
    _fields_ = [("start", C.c_int32), ("len", C.c_int32), ("energy", C.c_int32), ("printed", C.c_int32)]


FOLD_LINE_DTYPE = np.dtype([("start", "<i4"), ("len", "<i4"), ("energy", "<i4"), ("printed", "<i4")])

_lib = None


def load_library():
    """Load libmirprefer.so; raises MirpError loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MirpError("libmirprefer.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback for the product path)")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.mirp_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.mirp_create.restype = C.c_int
    lib.mirp_destroy.argtypes = [vp]
    lib.mirp_destroy.restype = None
    lib.mirp_last_error.argtypes = [vp]
    lib.mirp_last_error.restype = C.c_char_p
    lib.mirp_free.argtypes = [vp]
    lib.mirp_free.restype = None
    lib.mirp_abi_version.argtypes = []
    lib.mirp_abi_version.restype = C.c_int
    lib.mirp_fold_batch.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32,
                                    C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(vp),
                                    C.POINTER(vp), C.POINTER(vp)]
    lib.mirp_fold_batch.restype = C.c_int
    _lib = lib
    return lib


def _copy_out(lib, ptr, dtype, count):
    if count == 0:
        arr = np.zeros(0, dtype=dtype)
    else:
        nbytes = int(count) * np.dtype(dtype).itemsize
        buf = (C.c_char * nbytes).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dtype, count=count).copy()
    lib.mirp_free(ptr)
    return arr


class Context:
    """One device context (maps onto the reference's process-per-piece model)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.mirp_create(int(device), C.byref(h))
        if rc != 0:
            raise MirpError("mirp_create(device=%d) failed with code %d (no usable GPU?)" % (device, rc))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.mirp_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise MirpError("%s failed (%d): %s" % (what, rc, self.lib.mirp_last_error(self.h).decode()))

    def fold_batch(self, seqs, span, max_lines=96):
        """RNALfold -L replacement. seqs: list of str/bytes. Returns a list (per sequence) of
        dicts {lines: [(ss, energy, start), ...] (printed only), mfe, status}."""
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        n = len(bs)
        offs = np.zeros(n + 1, dtype=np.int64)
        if n:
            offs[1:] = np.cumsum([len(b) for b in bs])
        blob = b"".join(bs)
        vp = C.c_void_p
        lines, ss, nl, mfe, st = vp(), vp(), vp(), vp(), vp()
        stride = C.c_int32()
        rc = self.lib.mirp_fold_batch(self.h, blob, offs.ctypes.data_as(C.POINTER(C.c_int64)), n, int(span), int(max_lines),
                                      C.byref(lines), C.byref(ss), C.byref(stride), C.byref(nl), C.byref(mfe), C.byref(st))
        self._check(rc, "mirp_fold_batch")
        stride = stride.value
        a_lines = _copy_out(self.lib, lines, FOLD_LINE_DTYPE, n * max_lines).reshape(n, max_lines)
        a_ss = _copy_out(self.lib, ss, np.uint8, n * max_lines * stride).reshape(n, max_lines, stride)
        a_nl = _copy_out(self.lib, nl, np.int32, n)
        a_mfe = _copy_out(self.lib, mfe, np.int32, n)
        a_st = _copy_out(self.lib, st, np.int32, n)
        out = []
        for w in range(n):
            ls = []
            for k in range(int(a_nl[w])):
                ln = a_lines[w, k]
                if not ln["printed"]:
                    continue
                ls.append((a_ss[w, k, :int(ln["len"])].tobytes().decode(), int(ln["energy"]), int(ln["start"])))
            out.append({"lines": ls, "mfe": int(a_mfe[w]), "status": int(a_st[w])})
        return out
