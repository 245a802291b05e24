"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so). TEST INFRASTRUCTURE ONLY:
used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

MAX_LINES = 256
MAX_SS = 512


class FoldLine(C.Structure):
    _fields_ = [("ss", C.c_char * MAX_SS), ("len", C.c_int), ("energy", C.c_int), ("start", C.c_int)]


class FoldResult(C.Structure):
    _fields_ = [("n_lines", C.c_int), ("overflow", C.c_int), ("mfe", C.c_int), ("lines", FoldLine * MAX_LINES)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.oracle_lfold.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(FoldResult)]
        lib.oracle_lfold.restype = C.c_int

    def lfold(self, seq, span):
        """Returns {'lines': [(ss, energy_dcal, start)], 'mfe': int} exactly as RNALfold -L prints."""
        b = seq.encode() if isinstance(seq, str) else bytes(seq)
        r = FoldResult()
        rc = self.lib.oracle_lfold(b, len(b), int(span), C.byref(r))
        if rc != 0:
            raise RuntimeError("oracle_lfold rc=%d" % rc)
        lines = [(r.lines[k].ss.decode(), r.lines[k].energy, r.lines[k].start) for k in range(r.n_lines)]
        return {"lines": lines, "mfe": r.mfe}


_inst = None


def load():
    global _inst
    if _inst is None:
        if not os.path.exists(LIB):
            build()
        _inst = Oracle(C.CDLL(LIB))
    return _inst
