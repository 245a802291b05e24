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
    lib.mirp_predict_batch.argtypes = [vp, vp, C.c_int32, vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int32, C.c_int32, vp, vp,
                                       C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.mirp_predict_batch.restype = C.c_int
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
        raw = self.fold_batch_raw(seqs, span, max_lines)
        out = []
        for w in range(len(seqs)):
            ls = []
            for k in range(int(raw["n_lines"][w])):
                ln = raw["lines"][w, k]
                if not ln["printed"]:
                    continue
                ls.append((raw["ss"][w, k, :int(ln["len"])].tobytes().decode(), int(ln["energy"]), int(ln["start"])))
            out.append({"lines": ls, "mfe": int(raw["mfe"][w]), "status": int(raw["status"][w])})
        return out

    def fold_batch_raw(self, seqs, span, max_lines=96):
        """As fold_batch but returns the C-ABI arrays: lines[n,max_lines], ss[n,max_lines,stride], n_lines, mfe, status."""
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
        return {"lines": a_lines, "ss": a_ss, "stride": stride, "max_lines": max_lines, "n_lines": a_nl, "mfe": a_mfe, "status": a_st}

    def predict_batch(self, windows, matures, alns, fold_raw, params):
        """filter_next_loci/check_loci replacement. windows/matures/alns: record arrays (records.py dtypes);
        fold_raw: output of fold_batch_raw for the same windows; params: (n_samples, min_mature_len, max_mature_len,
        allow_3nt, allow_no_star, minlen).  Returns (mirnas[n, MAX_MIRNA_PER_WINDOW], n_mirnas[n], status[n])."""
        from . import records
        windows = np.ascontiguousarray(windows, dtype=records.WINDOW_DTYPE)
        matures = np.ascontiguousarray(matures, dtype=records.MATURE_DTYPE)
        alns = np.ascontiguousarray(alns)
        lines = np.ascontiguousarray(fold_raw["lines"])
        ss = np.ascontiguousarray(fold_raw["ss"])
        nl = np.ascontiguousarray(fold_raw["n_lines"], dtype=np.int32)
        n = len(windows)
        pp = (C.c_int32 * 6)(*[int(x) for x in params])
        vp = C.c_void_p
        o, no, st = vp(), vp(), vp()
        rc = self.lib.mirp_predict_batch(self.h, windows.ctypes.data, n, matures.ctypes.data, len(matures), alns.ctypes.data, len(alns),
                                         lines.ctypes.data, ss.ctypes.data, int(fold_raw["stride"]), int(fold_raw["max_lines"]),
                                         nl.ctypes.data, pp, C.byref(o), C.byref(no), C.byref(st))
        self._check(rc, "mirp_predict_batch")
        a_o = _copy_out(self.lib, o, records.MIRNA_DTYPE, n * records.MAX_MIRNA_PER_WINDOW).reshape(n, records.MAX_MIRNA_PER_WINDOW)
        return a_o, _copy_out(self.lib, no, np.int32, n), _copy_out(self.lib, st, np.int32, n)
