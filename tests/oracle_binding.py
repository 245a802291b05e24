"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so). TEST INFRASTRUCTURE ONLY:
used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

MAX_LINES = 384
MAX_SS = 3200


class FoldLine(C.Structure):
    _fields_ = [("ss", C.c_char * MAX_SS), ("len", C.c_int), ("energy", C.c_int), ("start", C.c_int)]


class FoldResult(C.Structure):
    _fields_ = [("n_lines", C.c_int), ("overflow", C.c_int), ("mfe", C.c_int), ("lines", FoldLine * MAX_LINES)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


MAX_STRUCTS = 512
MAX_SAMPLES = 256


class Struct(C.Structure):
    _fields_ = [("norm_energy", C.c_double), ("fold_start", C.c_int32), ("sstype", C.c_int32), ("len", C.c_int32),
                ("ss", C.c_char * MAX_SS)]


class MatureStar(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("code", "star_s", "star_e", "fold_s", "fold_e", "prime5", "total_dots", "total_bps",
                                         "star_l0", "star_l1", "mat_l0", "mat_l1")]


class Expr(C.Structure):
    _fields_ = [("reads_pre", C.c_int32 * MAX_SAMPLES), ("reads_mature", C.c_int32 * MAX_SAMPLES), ("reads_star", C.c_int32 * MAX_SAMPLES),
                ("reads_antisense", C.c_int32 * MAX_SAMPLES), ("reads_isoform", C.c_int32 * MAX_SAMPLES),
                ("reads_inside", C.c_int32 * MAX_SAMPLES), ("bases_with_reads_start", C.c_int32 * MAX_SAMPLES),
                ("imperfect", (C.c_int32 * 3) * MAX_SAMPLES),
                ("total_this_strand", C.c_int64), ("total_anti", C.c_int64), ("total_mature", C.c_int64), ("total_isoform", C.c_int64),
                ("total_star", C.c_int64), ("total_star_perfect", C.c_int64), ("total_imperfect", C.c_int64 * 3),
                ("mature_star_distance", C.c_int32), ("has_imperfect_key", C.c_int32), ("imperfect_which", C.c_int32),
                ("imperfect_start", C.c_int32), ("imperfect_end", C.c_int32), ("max_imperfect", C.c_int64),
                ("ratio_total", C.c_double), ("ratio_both", C.c_double), ("ratio_iso", C.c_double),
                ("ratio_start", C.c_double * MAX_SAMPLES), ("exception", C.c_int32)]


class Mirna(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("window", "tid", "fold_s", "fold_e", "mat_s", "mat_e", "star_s", "star_e", "strand", "has_star",
                                         "ss_len")] + [("ss", C.c_char * MAX_SS), ("total_depth_mature", C.c_int64),
                                                       ("total_depth_star", C.c_int64)]


class PredictParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_samples", "min_mature_len", "max_mature_len", "allow_3nt", "allow_no_star", "minlen")]


MS_CODES = [None, "FAIL_STRUCTURE_MATCHED_BASES", "FAIL_STRUCTURE_MATURE_NOT_IN_FOLD_REGION", "FAIL_STRUCTURE_MATURE_NOT_IN_ONE_ARM",
            "FAIL_STRUCTURE_MATURE_MATCH_SMALL_THAN_14", "FAIL_STRUCTURE_MATURE_STAR_OVERLAP", "FAIL_STRUCTURE_STAR_OUT_OF_FOLD_REGION",
            "FAIL_STRUCTURE_STAR_NOT_IN_ONE_ARM", "FAIL_STRUCTURE_TOO_MANY_BULGE_OR_LOOP", "FAIL_STRUCTURE_MAX_BULGE_LARGE_THAN_2",
            "FAIL_STRUCTURE_TOTAL_LOOP_SIZE_LARGER_THAN_5", "FAIL_STRUCTURE_NUM_BULGE_MORE_THAN_2", "REFERENCE_EXCEPTION"]


def _take(lib, ptr, dtype, n):
    import numpy as np
    if n == 0 or not ptr:
        arr = np.zeros(0, dtype=dtype)
    else:
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        arr = np.frombuffer(buf, dtype=dtype, count=n).copy()
    if ptr:
        lib.oracle_free(ptr)
    return arr


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.oracle_lfold.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(FoldResult)]
        lib.oracle_lfold.restype = C.c_int
        lib.oracle_lfold185.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(FoldResult)]
        lib.oracle_lfold185.restype = C.c_int
        lib.oracle_lfold_text.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.oracle_lfold_text.restype = C.c_int
        lib.oracle_free_text.argtypes = [C.c_void_p]
        lib.oracle_free_text.restype = None
        lib.oracle_free.argtypes = [C.c_void_p]
        lib.oracle_free.restype = None
        vp, sz = C.c_void_p, C.c_size_t
        lib.oracle_coverage_peaks.argtypes = [vp, sz, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz)]
        lib.oracle_coverage_peaks.restype = C.c_int
        lib.oracle_make_windows.argtypes = [vp, sz, vp, sz, C.POINTER(C.c_char_p), vp, C.c_int, vp, C.c_int, C.c_int, C.c_double] + \
            [C.POINTER(vp), C.POINTER(sz)] * 5
        lib.oracle_make_windows.restype = C.c_int
        lib.oracle_structures.argtypes = [C.POINTER(FoldLine), C.c_int, C.c_int, C.POINTER(Struct), C.c_int]
        lib.oracle_structures.restype = C.c_int
        lib.oracle_maturestar.argtypes = [C.c_char_p] + [C.c_int] * 7 + [C.POINTER(MatureStar)]
        lib.oracle_maturestar.restype = C.c_int
        lib.oracle_expression.argtypes = [vp, sz] + [C.c_int] * 12 + [C.POINTER(Expr)]
        lib.oracle_expression.restype = C.c_int
        lib.oracle_check_loci.argtypes = [C.POINTER(Struct), C.c_int, vp, C.c_int, vp, vp, sz, C.POINTER(PredictParams), C.POINTER(Mirna), C.c_int]
        lib.oracle_check_loci.restype = C.c_int

    # ---- candidate stage
    def coverage_peaks(self, alns, contig_lens, cutoff, min_len=19):
        import numpy as np
        from mir_prefer_amd import records
        alns = np.ascontiguousarray(alns)
        cl = np.ascontiguousarray(contig_lens, dtype=np.int64)
        d, nd, p, npk = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        rc = self.lib.oracle_coverage_peaks(alns.ctypes.data, len(alns), cl.ctypes.data, len(cl), cutoff, min_len,
                                            C.byref(d), C.byref(nd), C.byref(p), C.byref(npk))
        assert rc == 0
        return _take(self.lib, d.value, records.DEPTH_DTYPE, nd.value), _take(self.lib, p.value, records.PEAK_DTYPE, npk.value)

    def make_windows(self, peaks, alns, contigs, contig_order, max_gap, precursor_len, min_mature_depth):
        import numpy as np
        from mir_prefer_amd import records
        peaks = np.ascontiguousarray(peaks); alns = np.ascontiguousarray(alns)
        cl = np.array([len(s) for _, s in contigs], dtype=np.int64)
        keep = [s.tobytes() for _, s in contigs]
        garr = (C.c_char_p * len(keep))(*keep)
        order = np.ascontiguousarray(contig_order, dtype=np.int32)
        vp, sz = C.c_void_p, C.c_size_t
        outs = [(vp(), sz()) for _ in range(5)]
        args = []
        for a, b in outs:
            args += [C.byref(a), C.byref(b)]
        rc = self.lib.oracle_make_windows(peaks.ctypes.data, len(peaks), alns.ctypes.data, len(alns), garr, cl.ctypes.data, len(cl),
                                          order.ctypes.data, max_gap, precursor_len, float(min_mature_depth), *args)
        assert rc == 0
        w = _take(self.lib, outs[0][0].value, records.WINDOW_DTYPE, outs[0][1].value)
        wp = _take(self.lib, outs[1][0].value, records.PEAK_DTYPE, outs[1][1].value)
        mt = _take(self.lib, outs[2][0].value, records.MATURE_DTYPE, outs[2][1].value)
        seq = _take(self.lib, outs[3][0].value, np.uint8, outs[3][1].value)
        loci = _take(self.lib, outs[4][0].value, records.LOCUS_DTYPE, outs[4][1].value)
        return {"windows": w, "wpeaks": wp, "matures": mt, "seq": seq, "loci": loci}

    # ---- predict stage
    def lfold_raw(self, seq, span):
        b = seq.encode() if isinstance(seq, str) else bytes(seq)
        r = FoldResult()
        rc = self.lib.oracle_lfold(b, len(b), int(span), C.byref(r))
        if rc != 0:
            raise RuntimeError("oracle_lfold rc=%d" % rc)
        return r

    def structures_from_lines(self, lines, minlen=55):
        """lines: [(ss, energy_dcal, start)] -> [(norm_energy, fold_start, ss, sstype)]"""
        arr = (FoldLine * max(1, len(lines)))()
        for k, (ss, e, st) in enumerate(lines):
            arr[k].ss = ss.encode(); arr[k].len = len(ss); arr[k].energy = e; arr[k].start = st
        out = (Struct * MAX_STRUCTS)()
        n = self.lib.oracle_structures(arr, len(lines), minlen, out, MAX_STRUCTS)
        return [(out[k].norm_energy, out[k].fold_start, out[k].ss.decode(), out[k].sstype) for k in range(n)]

    def duplex_code(self, mature, star):
        self.lib.oracle_duplex_code.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        return int(self.lib.oracle_duplex_code(mature.encode(), len(mature), star.encode(), len(star)))

    def maturestar(self, ss, m0, m1, foldstart, regionstart, regionend, strand):
        o = MatureStar()
        self.lib.oracle_maturestar(ss.encode(), len(ss), m0, m1, foldstart, regionstart, regionend, strand, C.byref(o))
        return o

    def expression(self, alns, n_samples, tid, ws, we, fold_s, fold_e, m0, m1, star_s, star_e, strand, allow_3nt):
        o = Expr()
        self.lib.oracle_expression(alns.ctypes.data, len(alns), n_samples, tid, ws, we, fold_s, fold_e, m0, m1, star_s, star_e, strand,
                                   1 if allow_3nt else 0, C.byref(o))
        return o

    def check_loci(self, structs, matures, window, alns, params):
        """structs: [(norm_energy, fold_start, ss, sstype)]; matures: MATURE_DTYPE array; window: WINDOW_DTYPE scalar (0-d array)"""
        import numpy as np
        st = (Struct * max(1, len(structs)))()
        for k, (ne, fs, ss, ty) in enumerate(structs):
            st[k].norm_energy = ne; st[k].fold_start = fs; st[k].sstype = ty; st[k].len = len(ss); st[k].ss = ss.encode()
        mat = np.ascontiguousarray(matures)
        w = np.ascontiguousarray(window)
        pp = PredictParams(*params)
        out = (Mirna * 64)()
        n = self.lib.oracle_check_loci(st, len(structs), mat.ctypes.data, len(mat), w.ctypes.data, alns.ctypes.data, len(alns), C.byref(pp), out, 64)
        return [out[k] for k in range(n)]

    def lfold(self, seq, span, model="vienna-2.1.2"):
        """Returns {'lines': [(ss, energy_dcal, start)], 'mfe': int} exactly as RNALfold -L prints (RNALfold 2.1.2 by default,
        model="vienna-1.8.5" for the Turner-1999 / dangles-1 flavour)."""
        b = seq.encode() if isinstance(seq, str) else bytes(seq)
        if min(len(b), int(span)) + 60 >= MAX_SS or len(b) > 1500:      # longer or more lines than the fixed-size result holds (PRECURSOR_LEN above 1000): the text form
            text, nl, mfe = C.c_void_p(), C.c_int(), C.c_int()
            rc = self.lib.oracle_lfold_text(b, len(b), int(span), 1 if model == "vienna-1.8.5" else 0, C.byref(text), C.byref(nl), C.byref(mfe))
            try:
                if rc != 0:
                    raise RuntimeError("oracle_lfold_text rc=%d" % rc)
                rows = C.string_at(text.value).decode().splitlines()
            finally:
                self.lib.oracle_free_text(text)
            assert len(rows) == nl.value
            return {"lines": [(x.split(" ")[0], int(x.split(" ")[1]), int(x.split(" ")[2])) for x in rows], "mfe": mfe.value}
        r = FoldResult()
        fn = self.lib.oracle_lfold185 if model == "vienna-1.8.5" else self.lib.oracle_lfold
        rc = fn(b, len(b), int(span), C.byref(r))
        if rc != 0:
            raise RuntimeError("oracle_lfold rc=%d" % rc)
        if r.overflow:
            raise RuntimeError("oracle_lfold: more than ORACLE_MAX_LINES structure lines")
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
