#!/usr/bin/env python3
"""Dev-only (build container): load the Python-2 reference /root/reference/miR_PREFeR.py under
Python 3 through in-memory shims (nothing of the reference is copied or modified) so that its
functions can be called and its pipeline run to generate golden vectors.  Recipe: SURVEY.md
Appendix E1.  Never runs on the GPU box; never imported by product code.

    g = load_reference()            # namespace dict with the reference's functions
    run_pipeline(["-L","-k","pipeline",cfg])
"""
import ast
import builtins
import os
import pickle
import queue
import string
import subprocess
import sys
import types
import warnings

REF = "/root/reference/miR_PREFeR.py"
ORA_BIN = os.environ.get("MIRP_ORACLE_BIN", "/tmp/ora/bin")


def _install_shims(folder="RNALfold212"):
    warnings.filterwarnings("ignore")
    shim_bin = os.path.join(ORA_BIN, "path_" + folder)
    os.makedirs(shim_bin, exist_ok=True)
    for name, target in (("samtools", "samtools"), ("RNALfold", folder)):
        link = os.path.join(shim_bin, name)
        if not os.path.exists(link):
            os.symlink(os.path.join(ORA_BIN, target), link)
    os.environ["PATH"] = shim_bin + ":" + os.environ["PATH"]
    sys.modules["Queue"] = queue

    class S(str):
        def decode(self, *a, **k):
            return self

    _Popen = subprocess.Popen

    class Popen(_Popen):
        def __init__(self, *a, **k):
            k.setdefault("universal_newlines", True)
            super().__init__(*a, **k)

        def communicate(self, *a, **k):
            o, e = super().communicate(*a, **k)
            return (S(o) if o is not None else o, S(e) if e is not None else e)

    subprocess.Popen = Popen

    class _BinView:
        def __init__(self, f):
            self.f = f

        def read(self, n=-1):
            return self.f.read(n).encode("latin1")

        def readline(self):
            return self.f.readline().encode("latin1")

    cp = types.ModuleType("cPickle")

    def _dump(obj, f, protocol=0):
        data = pickle.dumps(obj, 0)
        if "b" in getattr(f, "mode", ""):
            f.write(data)
        else:
            f.write(data.decode("latin1"))

    def _load(f):
        if "b" in getattr(f, "mode", ""):
            return pickle.load(f)
        return pickle.Unpickler(_BinView(f)).load()

    cp.dump = _dump
    cp.load = _load
    sys.modules["cPickle"] = cp
    string.maketrans = str.maketrans
    builtins.xrange = range


def _compile_reference():
    src = open(REF).read()
    tree = ast.parse(src)

    class Fix(ast.NodeTransformer):
        def visit_BinOp(self, n):
            self.generic_visit(n)
            if isinstance(n.op, ast.Div) and n.lineno in (1293, 3071, 3195):
                n.op = ast.FloorDiv()  # py2 int division
            return n

        def visit_Raise(self, n):
            if n.exc is not None and "StopIteration" in ast.dump(n.exc):
                return ast.copy_location(ast.Return(value=None), n)  # PEP 479
            return n

    tree = ast.fix_missing_locations(Fix().visit(tree))
    return compile(tree, "miR_PREFeR.py", "exec")


def load_reference(folder="RNALfold212"):
    _install_shims(folder)
    g = {"__name__": "mirprefer_reference", "__file__": REF}
    exec(_compile_reference(), g)
    return g


def run_pipeline(argv, folder="RNALfold212"):
    _install_shims(folder)
    old = sys.argv
    sys.argv = ["miR_PREFeR.py"] + list(argv)
    g = {"__name__": "__main__", "__file__": REF}
    try:
        exec(_compile_reference(), g)
    except SystemExit as e:
        if e.code not in (0, None):
            raise
    finally:
        sys.argv = old
    return g


if __name__ == "__main__":
    run_pipeline(sys.argv[1:], os.environ.get("MIRP_FOLDER", "RNALfold212"))
