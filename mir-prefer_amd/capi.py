"""ctypes binding of libmirprefer.so (include/mirprefer.h).

There is no CPU fallback: if the shared library is missing or no GPU is usable, the calls raise."""
import ctypes as C
import os

import numpy as np

from . import early
from .early import LIB_PATH, FastaData, SamData
ABI_VERSION = 7      # include/mirprefer.h as this binding was written against (mirp_abi_version of the library must match)


class MirpError(RuntimeError):
    pass


class FoldLine(C.Structure):
    _fields_ = [("start", C.c_int32), ("len", C.c_int32), ("energy", C.c_int32), ("printed", C.c_int32)]


class Region(C.Structure):
    _fields_ = [("tid", C.c_int32), ("start", C.c_int32), ("end", C.c_int32)]


def report_readmapping(loci, ss_list, alns, contig_arrays, sample_names, counts0):
    """Bodies of the per-locus read-mapping files (mirp_report_readmapping, host only): loci = int32 [n, 8] {tid, fold_s, fold_e, mat_s, mat_e,
    star_s, star_e, strand}, ss_list = n structure strings, contig_arrays[t] = uint8 bases of contig t (None / empty where not held),
    counts0 = int64 [n, n_samples] reads on the precursor.  -> list of n strings (the lines after the header line)."""
    lib = load_library()
    loci = np.ascontiguousarray(loci, dtype=np.int32).reshape(-1, 8)
    n = len(loci)
    if n == 0:
        return []
    alns = np.ascontiguousarray(alns)
    keep = [np.ascontiguousarray(a, dtype=np.uint8) if a is not None and len(a) else None for a in contig_arrays]
    ptrs = (C.c_void_p * len(keep))(*[a.ctypes.data if a is not None else None for a in keep])
    lens = np.array([len(a) if a is not None else 0 for a in keep], dtype=np.int64)
    ssb = b"".join(x.encode() + b"\0" for x in ss_list)
    snb = b"".join(x.encode() + b"\0" for x in sample_names)
    cnt = np.ascontiguousarray(counts0, dtype=np.int64)
    text, offs = C.c_void_p(), C.c_void_p()
    rc = lib.mirp_report_readmapping(loci.ctypes.data, n, ssb, alns.ctypes.data if len(alns) else None, len(alns), ptrs, lens.ctypes.data, len(keep), snb,
                                     len(sample_names), cnt.ctypes.data, C.byref(text), C.byref(offs))
    if rc != 0:
        raise MirpError("mirp_report_readmapping failed (%d): a locus lies on a contig this process does not hold" % rc)
    o = _copy_out(lib, offs, np.int64, n + 1)
    blob = C.string_at(text.value, int(o[-1]))
    lib.mirp_free(text)
    return [blob[o[k]:o[k + 1]].decode() for k in range(n)]


def write_reports(loci, contig_names, ss_list, pre_list, sample_names, counts, mirbase_form, paths):
    """The report files of the predict stage (mirp_write_reports, host only, native threads): loci = int32 [n, 10] {contig index, fold_s, fold_e, mat_s,
    mat_e, star_s, star_e, strand, star expressed, overhang code} in final order, ss_list / pre_list = n structure / forward-strand precursor strings,
    counts = int64 [n, n_samples, 4], mirbase_form = the four fixed pieces of the two miRBase search forms, paths = {"gff", "mature", "precursor",
    "ss", "csv", "html", "stat"} -> file name (missing = not written)."""
    lib = load_library()
    loci = np.ascontiguousarray(loci, dtype=np.int32).reshape(-1, 10)
    n = len(loci)
    cnt = np.ascontiguousarray(counts, dtype=np.int64).reshape(n, len(sample_names), 4)
    blob = lambda xs: b"".join(x.encode() + b"\0" for x in xs)
    p = lambda k: paths[k].encode() if paths.get(k) else None
    err = C.create_string_buffer(512)
    fn = lib.mirp_write_reports
    fn.restype = C.c_int
    fn.argtypes = [C.c_int64, C.c_void_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_char_p] + [C.c_char_p] * 7 + \
                  [C.c_char_p, C.c_size_t]
    rc = fn(n, loci.ctypes.data if n else None, blob(contig_names), len(contig_names), blob(ss_list), blob(pre_list), blob(sample_names), len(sample_names),
            cnt.ctypes.data if cnt.size else None, blob(mirbase_form), p("gff"), p("mature"), p("precursor"), p("ss"), p("csv"), p("html"), p("stat"), err, 512)
    if rc != 0:
        raise MirpError("%s (%d)" % (err.value.decode(), rc))


def write_result_reports(result, text, contig_names, contig_arrays, alns, sample_names, mirbase_form, outdir, prefix):
    """The tail of the predict stage in one native call (mirp_write_result_reports, host only): result = MIRNA_DTYPE records and text = their structure
    rows as Context.predict_raw returns them, contig_arrays[t] = uint8 bases of contig t (None / empty where not held).  Writes readmapping/ and the
    seven report files under outdir.  -> (records in list order after the mature/star swap, order[n] = input index of list position i,
    counts int64 [n, n_samples, 4])."""
    lib = load_library()
    result = np.ascontiguousarray(result)
    n = len(result)
    text = np.ascontiguousarray(text, dtype=np.uint8).reshape(n, -1) if n else np.zeros((0, 1), np.uint8)
    alns = np.ascontiguousarray(alns)
    keep = [np.ascontiguousarray(a, dtype=np.uint8) if a is not None and len(a) else None for a in contig_arrays]
    ptrs = (C.c_void_p * max(len(keep), 1))(*[a.ctypes.data if a is not None else None for a in keep])
    lens = np.array([len(a) if a is not None else 0 for a in keep] or [0], dtype=np.int64)
    blob = lambda xs: b"".join(x.encode() + b"\0" for x in xs)
    order = np.zeros(max(n, 1), dtype=np.int32)
    out = np.zeros(max(n, 1), dtype=result.dtype)
    counts = np.zeros((max(n, 1), len(sample_names), 4), dtype=np.int64)
    err = C.create_string_buffer(512)
    fn = lib.mirp_write_result_reports
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_int32, C.c_char_p,
                   C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    rc = fn(result.ctypes.data if n else None, n, text.ctypes.data if n else None, int(text.shape[1]), blob(contig_names), len(contig_names), ptrs, lens.ctypes.data,
            alns.ctypes.data if len(alns) else None, len(alns), blob(sample_names), len(sample_names), blob(mirbase_form), str(outdir).encode(), str(prefix).encode(),
            order.ctypes.data, out.ctypes.data, counts.ctypes.data, err, 512)
    if rc != 0:
        raise MirpError("%s (%d)" % (err.value.decode(), rc))
    return out[:n], order[:n], counts[:n]


def write_files(paths, texts, n_threads=1):
    """mirp_write_files: file paths[k] <- texts[k] (str), written natively, outside the interpreter lock."""
    lib = load_library()
    n = len(paths)
    if n == 0:
        return
    enc = [t.encode() for t in texts]
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(e) for e in enc], out=offs[1:])
    err = C.create_string_buffer(512)
    fn = lib.mirp_write_files
    fn.restype = C.c_int
    fn.argtypes = [C.c_int64, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_char_p, C.c_size_t]
    rc = fn(n, b"".join(p.encode() + b"\0" for p in paths), b"".join(enc), offs.ctypes.data, int(n_threads), err, 512)
    if rc != 0:
        raise MirpError("%s (%d)" % (err.value.decode(), rc))


def read_fasta(path, want=None):
    """Native FASTA reader (mirp_read_fasta): -> list of (name, uint8 array) in file order; with `want` (names) only those sequences are
    materialised, the others come back as None."""
    lib = load_library()
    got = early.take_fasta(path) if not want else None      # the CLI may have started this read before the heavy imports (early.py)
    if got is not None:
        rc, d, msg = got
        if rc != 0:
            raise ValueError(msg)
    else:
        d = FastaData()
        err = C.create_string_buffer(512)
        nw = len(want) if want else 0
        arr = (C.c_char_p * max(nw, 1))(*[str(w).encode() for w in (want or [])])
        if lib.mirp_read_fasta(str(path).encode(), arr, nw, C.byref(d), err, 512) != 0:
            raise ValueError(err.value.decode())
    keep = got is not None          # the CLI's early read: the process is short-lived, the arrays below are views of the library's buffer (no 119 MB copy
    try:                            # at config[2]); the buffer is never handed back -- _FASTA_KEEP holds the descriptor until the process ends
        raw = np.frombuffer((C.c_char * d.n_bytes).from_address(d.seq), dtype=np.uint8, count=d.n_bytes) if d.n_bytes else np.zeros(0, np.uint8)
        blob = raw if keep else raw.copy()
        out, off, o = [], 0, 0
        for k in range(d.n_contigs):
            nm = C.string_at(d.names + off)
            off += len(nm) + 1
            L = d.len[k]
            if L < 0:
                out.append((nm.decode(), None))
            else:
                out.append((nm.decode(), blob[o:o + L])); o += L
    finally:
        if keep:
            _FASTA_KEEP.append(d)
        else:
            lib.mirp_free_fasta_data(C.byref(d))
    return out


_FASTA_KEEP = []


class _SamHold:
    """Owns the buffers of one MirpSamData until the record arrays that view them are gone (mirp_free_sam_data then)."""

    def __init__(self, lib, d):
        self.lib, self.d = lib, d

    def __del__(self):
        try:
            self.lib.mirp_free_sam_data(C.byref(self.d))
        except Exception:
            pass


def _unpack_sam_data(lib, d):
    """-> (contig names, lengths, sample names, alns, segs).  The record arrays are VIEWS of the library's buffers (no copy of 16 bytes x 10^7 records:
    the copy was half of the ingest leg's wall-clock); the buffers are released when the last array viewing them is collected."""
    from .synth import ALN_DTYPE
    hold = _SamHold(lib, d)

    def names(ptr, n):
        out, off = [], 0
        for _ in range(n):
            s = C.string_at(ptr + off)
            out.append(s.decode()); off += len(s) + 1
        return out
    cn = names(d.contig_names, d.n_contigs)
    sn = names(d.sample_names, d.n_samples)
    lens = np.array([d.contig_len[k] for k in range(d.n_contigs)], dtype=np.int64)

    def recs(ptr, n):
        if not n:
            return np.zeros(0, dtype=ALN_DTYPE)
        buf = (C.c_char * (n * 16)).from_address(ptr)
        buf._hold = hold          # the array's base keeps the owner alive
        return np.frombuffer(buf, dtype=ALN_DTYPE, count=n)
    alns, segs = recs(d.alns, d.n_alns), recs(d.segs, d.n_segs)
    return cn, lens, sn, alns, segs


def ingest_sams(paths, n_threads=0, with_segments=False):
    """Native multi-threaded SAM ingest, host sort (mirp_ingest_sams). -> (contig_names, contig_lens, sample_names, alns[, segs])."""
    lib = load_library()
    arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
    d = SamData()
    err = C.create_string_buffer(512)
    if lib.mirp_ingest_sams(arr, len(paths), int(n_threads), C.byref(d), err, 512) != 0:
        raise ValueError(err.value.decode())
    out = _unpack_sam_data(lib, d)
    return out if with_segments else out[:4]


FOLD_LINE_DTYPE = np.dtype([("start", "<i4"), ("len", "<i4"), ("energy", "<i4"), ("printed", "<i4")])


def fold_window_lines(raw, w):
    """(lines, ss) record arrays of window w of a Context.get_fold() result: the side buffer for a window that needed more
    lines than the main buffers hold, its slot in the main buffers otherwise."""
    ov = raw.get("overflow")
    if ov and int(w) in ov:
        return ov[int(w)]
    return raw["lines"][w], raw["ss"][w]

_lib = None


def load_library():
    """Load libmirprefer.so; raises MirpError loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MirpError("libmirprefer.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback for the product path)")
    lib = early.cdll()
    lib.mirp_abi_version.argtypes = []
    lib.mirp_abi_version.restype = C.c_int
    if lib.mirp_abi_version() != ABI_VERSION:      # a stale build (or a stale MIRP_LIB variant) would be called through the wrong signatures
        raise MirpError("%s has ABI version %d, this binding needs %d: rebuild it (make -C mir-prefer_amd/csrc)" % (LIB_PATH, lib.mirp_abi_version(), ABI_VERSION))
    if os.environ.get("MIRP_LIB"):
        import sys
        sys.stderr.write("[mirp] using the library named by MIRP_LIB: %s\n" % LIB_PATH)
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
    lib.mirp_fold_batch_summary.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.mirp_fold_batch_summary.restype = C.c_int
    lib.mirp_predict_batch.argtypes = [vp, vp, C.c_int32, vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int32, C.c_int32, vp, vp,
                                       C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.mirp_predict_batch.restype = C.c_int
    lib.mirp_predict_batch_reasons.argtypes = [vp, vp, C.c_int32, vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int32, C.c_int32, vp, vp,
                                               C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.mirp_predict_batch_reasons.restype = C.c_int
    i64p, i32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    lib.mirp_load_genome.argtypes = [vp, C.c_int32, vp, vp]
    lib.mirp_load_alignments.argtypes = [vp, vp, C.c_int64]
    lib.mirp_candidate.argtypes = [vp, vp, vp, i64p, i64p, i64p]
    lib.mirp_get_depth.argtypes = [vp, C.POINTER(vp), i64p]
    lib.mirp_get_peaks.argtypes = [vp, C.POINTER(vp), i64p]
    lib.mirp_get_loci.argtypes = [vp, C.POINTER(vp), i64p, C.POINTER(vp), i64p]
    lib.mirp_get_windows.argtypes = [vp, C.POINTER(vp), i64p, C.POINTER(vp), i64p, C.POINTER(vp), i64p, C.POINTER(vp), i64p]
    lib.mirp_fold.argtypes = [vp, C.c_int32, C.c_int32]
    lib.mirp_set_fold_model.argtypes = [vp, C.c_int32]
    lib.mirp_set_contig_shard.argtypes = [vp, C.c_int32]
    lib.mirp_set_contig_shard.restype = C.c_int
    lib.mirp_set_fold_model.restype = C.c_int
    lib.mirp_get_fold.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i32p, i32p, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.mirp_predict.argtypes = [vp, vp, C.POINTER(vp), i64p, C.POINTER(vp), i32p, C.POINTER(vp), C.POINTER(vp), i64p]
    lib.mirp_get_fold_overflow.argtypes = [vp, C.POINTER(vp), i64p, C.POINTER(vp), C.POINTER(vp), i32p, i32p, C.POINTER(vp)]
    lib.mirp_get_fold_overflow.restype = C.c_int
    lib.mirp_predict_reasons.argtypes = [vp, vp, C.POINTER(vp), i64p, i32p]
    lib.mirp_predict_reasons.restype = C.c_int
    lib.mirp_last_timings.argtypes = [vp, C.POINTER(C.c_double)]
    lib.mirp_last_fold_fallbacks.argtypes = [vp]
    lib.mirp_last_fold_fallbacks.restype = C.c_int64
    lib.mirp_get_window_readtable.argtypes = [vp, C.POINTER(vp), i32p, i64p]
    lib.mirp_get_window_readtable.restype = C.c_int
    lib.mirp_last_fold_kernel_ms.argtypes = [vp, C.POINTER(C.c_double)]
    lib.mirp_last_fold_kernel_ms.restype = C.c_int
    lib.mirp_microbench.argtypes = [vp, C.POINTER(C.c_double)]
    lib.mirp_microbench.restype = C.c_int
    lib.mirp_last_fold_overflow.argtypes = [vp]
    lib.mirp_last_fold_overflow.restype = C.c_int64
    lib.mirp_last_coverage_fused.argtypes = [vp]
    lib.mirp_last_coverage_fused.restype = C.c_int
    lib.mirp_set_coverage_path.argtypes = [vp, C.c_int32]
    lib.mirp_set_coverage_path.restype = C.c_int
    lib.mirp_limit_windows.argtypes = [vp, C.c_int64]
    lib.mirp_limit_windows.restype = C.c_int
    lib.mirp_exchange_bytes.argtypes = [vp, vp, vp, C.POINTER(vp), vp]
    lib.mirp_exchange_bytes.restype = C.c_int
    lib.mirp_set_fold_split_path.argtypes = [vp, C.c_int32]
    lib.mirp_set_fold_split_path.restype = C.c_int
    lib.mirp_last_fold_dense.argtypes = [vp]
    lib.mirp_last_fold_dense.restype = C.c_int64
    lib.mirp_write_fold_text.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.mirp_write_fold_text.restype = C.c_int
    lib.mirp_write_fold_text_async.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.mirp_write_fold_text_async.restype = C.c_int
    lib.mirp_wait_text.argtypes = [vp]
    lib.mirp_wait_text.restype = C.c_int
    for f in ("mirp_write_depth_text", "mirp_write_window_fasta"):
        getattr(lib, f).argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int32]
        getattr(lib, f).restype = C.c_int
    lib.mirp_get_fold_summary.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i64p]
    lib.mirp_get_fold_summary.restype = C.c_int
    for f in ("mirp_load_genome", "mirp_load_alignments", "mirp_candidate", "mirp_get_depth", "mirp_get_peaks", "mirp_get_loci",
              "mirp_get_windows", "mirp_fold", "mirp_get_fold", "mirp_predict", "mirp_last_timings"):
        getattr(lib, f).restype = C.c_int
    lib.mirp_ingest_sams.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.POINTER(SamData), C.c_char_p, C.c_size_t]
    lib.mirp_ingest_sams.restype = C.c_int
    lib.mirp_free_sam_data.argtypes = [C.POINTER(SamData)]
    lib.mirp_free_sam_data.restype = None
    lib.mirp_ingest_sams_gpu.argtypes = [vp, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, vp, C.c_int64, C.POINTER(SamData), C.POINTER(C.c_double)]
    lib.mirp_ingest_sams_gpu.restype = C.c_int
    lib.mirp_load_coverage_segments.argtypes = [vp, vp, C.c_int64]
    lib.mirp_load_coverage_segments.restype = C.c_int
    lib.mirp_ingest_sams_shard.argtypes = [vp, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, vp, C.c_int64, vp, C.POINTER(SamData), C.POINTER(C.c_double)]
    lib.mirp_ingest_sams_shard.restype = C.c_int
    lib.mirp_report_readmapping.argtypes = [vp, C.c_int64, C.c_char_p, vp, C.c_int64, vp, vp, C.c_int32, C.c_char_p, C.c_int32, vp, C.POINTER(vp), C.POINTER(vp)]
    lib.mirp_report_readmapping.restype = C.c_int
    lib.mirp_read_fasta.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(FastaData), C.c_char_p, C.c_size_t]
    lib.mirp_read_fasta.restype = C.c_int
    lib.mirp_free_fasta_data.argtypes = [C.POINTER(FastaData)]
    lib.mirp_free_fasta_data.restype = None
    lib.mirp_dist_unique_id.argtypes = [vp]
    lib.mirp_dist_unique_id.restype = C.c_int
    lib.mirp_dist_init.argtypes = [vp, vp, C.c_int32, C.c_int32]
    lib.mirp_dist_init.restype = C.c_int
    lib.mirp_dist_init_local.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32]
    lib.mirp_dist_init_local.restype = C.c_int
    for f in ("mirp_dist_finalize", "mirp_dist_barrier", "mirp_dist_rank", "mirp_dist_world"):
        getattr(lib, f).argtypes = [vp]
        getattr(lib, f).restype = C.c_int
    lib.mirp_dist_comm_info.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.mirp_dist_comm_info.restype = C.c_int
    lib.mirp_dist_allreduce_sum.argtypes = [vp, vp, C.c_int32]
    lib.mirp_dist_allreduce_sum.restype = C.c_int
    lib.mirp_gather_loci.argtypes = [vp, C.c_int32, C.POINTER(vp), i64p, C.POINTER(vp), i32p]
    lib.mirp_gather_loci.restype = C.c_int
    lib.mirp_gather_records.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int32, C.POINTER(vp), i64p]
    lib.mirp_gather_records.restype = C.c_int
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
        got = early.take_context(int(device))      # the CLI may have started the context before the heavy imports (early.py)
        if got is not None:
            rc, h = got
        else:
            h = C.c_void_p()
            rc = self.lib.mirp_create(int(device), C.byref(h))
        if rc != 0:
            raise MirpError("mirp_create(device=%d) failed with code %d (no usable GPU?)" % (device, rc))
        self.h = h
        self.device = int(device)

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
            for k in range(min(int(raw["n_lines"][w]), max_lines)):      # a window over the capacity (status 1) reports the count it needs
                ln = raw["lines"][w, k]
                if not ln["printed"]:
                    continue
                ls.append((raw["ss"][w, k, :int(ln["len"])].tobytes().decode(), int(ln["energy"]), int(ln["start"])))
            out.append({"lines": ls, "mfe": int(raw["mfe"][w]), "status": int(raw["status"][w])})
        return out

    def fold_batch_summary(self, blob, offsets, span, max_lines=96):
        """Folds the sequences blob[offsets[k]:offsets[k+1]] and returns only (n_lines, mfe, status) per sequence: nothing of the structure text
        is copied back (mirp_fold_batch_summary).  blob: bytes / uint8 array, offsets: int64 array of n + 1 entries."""
        blob = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, (bytes, bytearray)) else blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offs) - 1
        vp = C.c_void_p
        nl, mfe, st = vp(), vp(), vp()
        self._check(self.lib.mirp_fold_batch_summary(self.h, C.cast(blob.ctypes.data, C.c_char_p), offs.ctypes.data_as(C.POINTER(C.c_int64)), n, int(span),
                                                     int(max_lines), C.byref(nl), C.byref(mfe), C.byref(st)), "mirp_fold_batch_summary")
        return _copy_out(self.lib, nl, np.int32, n), _copy_out(self.lib, mfe, np.int32, n), _copy_out(self.lib, st, np.int32, n)

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

    def predict_batch_reasons(self, windows, matures, alns, fold_raw, params):
        """predict_batch plus the -d records (mirp_predict_batch_reasons): -> (mirnas, n_mirnas, status, reasons[n_records, stride]); the layout
        of a record is documented at mirp_predict_reasons in include/mirprefer.h."""
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
        o, no, st, rr = vp(), vp(), vp(), vp()
        nr, rs = C.c_int64(), C.c_int32()
        rc = self.lib.mirp_predict_batch_reasons(self.h, windows.ctypes.data, n, matures.ctypes.data, len(matures), alns.ctypes.data if len(alns) else None, len(alns),
                                                 lines.ctypes.data, ss.ctypes.data, int(fold_raw["stride"]), int(fold_raw["max_lines"]),
                                                 nl.ctypes.data, pp, C.byref(o), C.byref(no), C.byref(st), C.byref(rr), C.byref(nr), C.byref(rs))
        self._check(rc, "mirp_predict_batch_reasons")
        a_o = _copy_out(self.lib, o, records.MIRNA_DTYPE, n * records.MAX_MIRNA_PER_WINDOW).reshape(n, records.MAX_MIRNA_PER_WINDOW)
        rec = _copy_out(self.lib, rr, np.int32, nr.value * rs.value).reshape(nr.value, max(rs.value, 1)) if rr.value else np.zeros((0, max(rs.value, 1)), np.int32)
        return a_o, _copy_out(self.lib, no, np.int32, n), _copy_out(self.lib, st, np.int32, n), rec

    # ---- device-resident pipeline -------------------------------------------------------------
    def load_genome(self, contigs):
        """contigs: list of (name, uint8 array) in @SQ order."""
        lens = np.array([len(s) for _, s in contigs], dtype=np.int64)
        arrs = [s for _, s in contigs]
        base = arrs[0].base if arrs and isinstance(arrs[0], np.ndarray) else None
        if (isinstance(base, np.ndarray) and base.ndim == 1 and base.dtype == np.uint8 and base.flags.c_contiguous and len(base) == int(lens.sum()) and
                all(isinstance(a, np.ndarray) and a.dtype == np.uint8 for a in arrs) and arrs[0].ctypes.data == base.ctypes.data and
                all(a.ctypes.data + len(a) == b.ctypes.data for a, b in zip(arrs, arrs[1:]))):
            blob = base          # the contigs are consecutive slices of one buffer (capi.read_fasta): upload it as it is
        else:
            blob = np.concatenate(arrs) if contigs else np.zeros(0, np.uint8)
            blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._check(self.lib.mirp_load_genome(self.h, len(contigs), lens.ctypes.data, blob.ctypes.data), "mirp_load_genome")

    def load_alignments(self, alns):
        alns = np.ascontiguousarray(alns)
        assert alns.dtype.itemsize == 16
        self._check(self.lib.mirp_load_alignments(self.h, alns.ctypes.data, len(alns)), "mirp_load_alignments")

    def load_coverage_segments(self, segs):
        """Coverage segments of gapped alignments (see mirp_load_coverage_segments); call after load_alignments."""
        segs = np.ascontiguousarray(segs)
        assert segs.dtype.itemsize == 16
        self._check(self.lib.mirp_load_coverage_segments(self.h, segs.ctypes.data, len(segs)), "mirp_load_coverage_segments")

    def ingest_sams(self, paths, regions=None, n_threads=0):
        """SAM files -> sorted records with the device doing the record work (mirp_ingest_sams_gpu): host threads tokenize, the GPU filters by
        the keep regions [(tid, start0, end0), ...] like `samtools view -L` and sorts stably by (tid, pos).  The records stay resident as this
        context's alignments.  -> (contig_names, contig_lens, sample_names, alns, segs, seconds{tokenize, upload_filter, sort, download})."""
        arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
        nreg = len(regions) if regions else 0
        reg = (Region * max(nreg, 1))()
        for k in range(nreg):
            reg[k].tid, reg[k].start, reg[k].end = int(regions[k][0]), int(regions[k][1]), int(regions[k][2])
        d = SamData()
        sec = (C.c_double * 4)()
        import time
        t0 = time.time()
        rc = self.lib.mirp_ingest_sams_gpu(self.h, arr, len(paths), int(n_threads), C.cast(reg, C.c_void_p), nreg, C.byref(d), sec)
        t1 = time.time()
        if rc != 0:
            raise ValueError(self.lib.mirp_last_error(self.h).decode())
        cn, lens, sn, alns, segs = _unpack_sam_data(self.lib, d)
        # native_other_s: what the library call spends outside its four phases (opening / mapping the files, the host buffers of the result);
        # host_copy_s: the records copied out of the library's buffer into numpy arrays and the buffer released
        return cn, lens, sn, alns, segs, {"tokenize_s": sec[0], "upload_filter_s": sec[1], "sort_s": sec[2], "download_s": sec[3],
                                          "native_other_s": max(0.0, (t1 - t0) - sum(sec)), "host_copy_s": time.time() - t1}

    def ingest_tokenized(self, paths, regions=None):
        """The device half of ingest_sams for files whose tokenizer run the CLI started before its heavy imports (early.start_ingest): keep-region filter +
        stable radix sort (mirp_ingest_tokenized_gpu).  Same return as ingest_sams."""
        got = early.take_ingest(paths)
        if got is None:
            raise MirpError("ingest_tokenized: no tokenizer run was started for these files")
        rc, tok, msg = got
        if rc != 0:
            raise ValueError(msg)
        nreg = len(regions) if regions else 0
        reg = (Region * max(nreg, 1))()
        for k in range(nreg):
            reg[k].tid, reg[k].start, reg[k].end = int(regions[k][0]), int(regions[k][1]), int(regions[k][2])
        d = SamData()
        sec = (C.c_double * 4)()
        fn = self.lib.mirp_ingest_tokenized_gpu
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(SamData), C.POINTER(C.c_double)]
        fn.restype = C.c_int
        if fn(self.h, tok, C.cast(reg, C.c_void_p), nreg, C.byref(d), sec) != 0:
            raise ValueError(self.lib.mirp_last_error(self.h).decode())
        cn, lens, sn, alns, segs = _unpack_sam_data(self.lib, d)
        return cn, lens, sn, alns, segs, {"tokenize_s": 0.0, "upload_filter_s": sec[1], "sort_s": sec[2], "download_s": sec[3]}

    def ingest_sams_shard(self, paths, owner_of_tid, regions=None, n_threads=0):
        """Sharded ingest (mirp_ingest_sams_shard): this rank tokenizes its byte range of every file, records travel to the rank that owns their
        contig over RCCL (dist_init first), this rank filters / sorts / keeps its own.  Same return as ingest_sams, records of this rank's contigs."""
        arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
        nreg = len(regions) if regions else 0
        reg = (Region * max(nreg, 1))()
        for k in range(nreg):
            reg[k].tid, reg[k].start, reg[k].end = int(regions[k][0]), int(regions[k][1]), int(regions[k][2])
        own = np.ascontiguousarray(owner_of_tid, dtype=np.int32) if owner_of_tid is not None else None
        d = SamData()
        sec = (C.c_double * 4)()
        rc = self.lib.mirp_ingest_sams_shard(self.h, arr, len(paths), int(n_threads), C.cast(reg, C.c_void_p), nreg,
                                             own.ctypes.data if own is not None else None, C.byref(d), sec)
        if rc != 0:
            raise ValueError(self.lib.mirp_last_error(self.h).decode())
        cn, lens, sn, alns, segs = _unpack_sam_data(self.lib, d)
        return cn, lens, sn, alns, segs, {"tokenize_s": sec[0], "exchange_upload_filter_s": sec[1], "sort_s": sec[2], "download_s": sec[3]}

    # ---- multi-GPU: RCCL communicator of this context (one process per GPU) -------------------
    def dist_unique_id(self):
        """128 bytes of ncclGetUniqueId: rank 0 calls this and hands the bytes to every rank's dist_init."""
        buf = (C.c_uint8 * 128)()
        if self.lib.mirp_dist_unique_id(buf) != 0:
            raise MirpError("mirp_dist_unique_id failed (librccl.so.1 not loadable?)")
        return bytes(buf)

    def dist_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._check(self.lib.mirp_dist_init(self.h, buf, int(rank), int(world)), "mirp_dist_init")

    def dist_init_local(self, directory, rank, world):
        """Ranks that share one GPU: exchanges staged through files in `directory` (mirp_dist_init_local)."""
        self._check(self.lib.mirp_dist_init_local(self.h, str(directory).encode(), int(rank), int(world)), "mirp_dist_init_local")

    def dist_finalize(self):
        self._check(self.lib.mirp_dist_finalize(self.h), "mirp_dist_finalize")

    def dist_world(self):
        return int(self.lib.mirp_dist_world(self.h))

    def dist_comm_info(self):
        """{ncclCommCount, ncclCommUserRank, ncclCommCuDevice} of the context's RCCL communicator; -1s without one."""
        v = (C.c_int32 * 3)()
        self._check(self.lib.mirp_dist_comm_info(self.h, v), "mirp_dist_comm_info")
        return {"comm_count": int(v[0]), "comm_rank": int(v[1]), "comm_device": int(v[2])}

    def dist_barrier(self):
        self._check(self.lib.mirp_dist_barrier(self.h), "mirp_dist_barrier")

    def dist_allreduce_sum(self, values):
        v = np.ascontiguousarray(values, dtype=np.int64).copy()
        self._check(self.lib.mirp_dist_allreduce_sum(self.h, v.ctypes.data, len(v)), "mirp_dist_allreduce_sum")
        return v

    def gather_loci(self, dst=0):
        """The result of the last predict() of every rank on rank dst (rank order): {"result": MIRNA records, "ss": [str]}; empty elsewhere."""
        from . import records
        r, t = C.c_void_p(), C.c_void_p()
        n, stride = C.c_int64(), C.c_int32()
        self._check(self.lib.mirp_gather_loci(self.h, int(dst), C.byref(r), C.byref(n), C.byref(t), C.byref(stride)), "mirp_gather_loci")
        res = _copy_out(self.lib, r, records.MIRNA_DTYPE, n.value) if r.value else np.zeros(0, dtype=records.MIRNA_DTYPE)
        txt = _copy_out(self.lib, t, np.uint8, n.value * stride.value).reshape(n.value, stride.value) if t.value else np.zeros((0, max(stride.value, 1)), np.uint8)
        ss = [txt[k, :res[k]["ss_len"]].tobytes().decode() for k in range(n.value)]
        return {"result": res, "ss": ss}

    def gather_records(self, rec, dst=0):
        """Fixed-size records (a C-contiguous numpy array, first axis = records) of every rank on rank dst, in rank order; None elsewhere."""
        rec = np.ascontiguousarray(rec)
        rb = int(rec.dtype.itemsize * (rec.size // max(len(rec), 1))) if len(rec) else int(rec.dtype.itemsize * int(np.prod(rec.shape[1:], dtype=np.int64)))
        o, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.mirp_gather_records(self.h, rec.ctypes.data if len(rec) else None, len(rec), max(rb, 1), int(dst), C.byref(o), C.byref(n)),
                    "mirp_gather_records")
        if not o.value:
            return None
        flat = _copy_out(self.lib, o, np.uint8, n.value * max(rb, 1))
        return flat.view(rec.dtype).reshape((n.value,) + rec.shape[1:])

    def candidate(self, cutoff, max_gap, precursor_len, contig_order, min_peak_len=19):
        pp = (C.c_int32 * 4)(int(cutoff), int(min_peak_len), int(max_gap), int(precursor_len))
        order = np.ascontiguousarray(contig_order, dtype=np.int32)
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.mirp_candidate(self.h, pp, order.ctypes.data, C.byref(a), C.byref(b), C.byref(c)), "mirp_candidate")
        self._n_windows = c.value
        return a.value, b.value, c.value

    def get_depth(self):
        from . import records
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.mirp_get_depth(self.h, C.byref(p), C.byref(n)), "mirp_get_depth")
        return _copy_out(self.lib, p, records.DEPTH_DTYPE, n.value)

    def get_peaks(self):
        from . import records
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.mirp_get_peaks(self.h, C.byref(p), C.byref(n)), "mirp_get_peaks")
        return _copy_out(self.lib, p, records.PEAK_DTYPE, n.value)

    def get_loci(self):
        from . import records
        p, n, q, m = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        self._check(self.lib.mirp_get_loci(self.h, C.byref(p), C.byref(n), C.byref(q), C.byref(m)), "mirp_get_loci")
        return _copy_out(self.lib, p, records.LOCUS_DTYPE, n.value), _copy_out(self.lib, q, records.PEAK_DTYPE, m.value)

    def get_windows(self):
        from . import records
        vp = C.c_void_p
        w, nw, pk, npk, mt, nmt, sq, nsq = vp(), C.c_int64(), vp(), C.c_int64(), vp(), C.c_int64(), vp(), C.c_int64()
        self._check(self.lib.mirp_get_windows(self.h, C.byref(w), C.byref(nw), C.byref(pk), C.byref(npk), C.byref(mt), C.byref(nmt),
                                              C.byref(sq), C.byref(nsq)), "mirp_get_windows")
        return {"windows": _copy_out(self.lib, w, records.WINDOW_DTYPE, nw.value), "wpeaks": _copy_out(self.lib, pk, records.PEAK_DTYPE, npk.value),
                "matures": _copy_out(self.lib, mt, records.MATURE_DTYPE, nmt.value), "seq": _copy_out(self.lib, sq, np.uint8, nsq.value)}

    def get_window_readtable(self):
        """int32 [n_windows, width, 3]: {length of the most abundant read, its depth, total depth} per start position ws + x of the window's strand."""
        t, w, n = C.c_void_p(), C.c_int32(), C.c_int64()
        self._check(self.lib.mirp_get_window_readtable(self.h, C.byref(t), C.byref(w), C.byref(n)), "mirp_get_window_readtable")
        return _copy_out(self.lib, t, np.int32, n.value * w.value * 3).reshape(n.value, w.value, 3)

    FOLD_MODELS = {"vienna-2.1.2": 0, "vienna-1.8.5": 1}

    def set_fold_model(self, model):
        """Which RNALfold the fold entry points reproduce: "vienna-2.1.2" (Turner-2004, dangles 2; default) or "vienna-1.8.5"
        (Turner-1999, dangles 1: the Linux binary the reference bundles)."""
        if model not in self.FOLD_MODELS:
            raise MirpError("unknown fold model %r (expected one of %s)" % (model, ", ".join(sorted(self.FOLD_MODELS))))
        self._check(self.lib.mirp_set_fold_model(self.h, self.FOLD_MODELS[model]), "mirp_set_fold_model")

    def set_contig_shard(self, preceded_by_coverage_elsewhere):
        """Contig sharding: see mirp_set_contig_shard in include/mirprefer.h."""
        self._check(self.lib.mirp_set_contig_shard(self.h, 1 if preceded_by_coverage_elsewhere else 0), "mirp_set_contig_shard")

    def fold(self, span, max_lines=96):
        self._check(self.lib.mirp_fold(self.h, int(span), int(max_lines)), "mirp_fold")

    def last_fold_fallbacks(self):
        return int(self.lib.mirp_last_fold_fallbacks(self.h))

    def last_fold_overflow(self):
        return int(self.lib.mirp_last_fold_overflow(self.h))

    def excl_scan(self, values):
        """Exclusive prefix sums of an int32 array on the device (mirp_excl_scan_i32, the scan of the candidate stage's compactions) -> int64 [n + 1]."""
        v = np.ascontiguousarray(values, dtype=np.int32)
        out = np.zeros(len(v) + 1, dtype=np.int64)
        fn = self.lib.mirp_excl_scan_i32
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        self._check(fn(self.h, v.ctypes.data if len(v) else None, len(v), out.ctypes.data), "mirp_excl_scan_i32")
        return out

    def limit_windows(self, n_keep):
        """Keeps the first n_keep windows of the last candidate() for the following stages (mirp_limit_windows)."""
        self._check(self.lib.mirp_limit_windows(self.h, int(n_keep)), "mirp_limit_windows")
        self._n_windows = int(n_keep)

    def exchange_bytes(self, blocks):
        """All-to-all of byte strings on the context's communicator: blocks[q] goes to rank q; returns the list of blocks received, by source rank."""
        W = len(blocks)
        cnt = np.array([len(b) for b in blocks], dtype=np.int64)
        blob = b"".join(blocks)
        recv, rcnt = C.c_void_p(), np.zeros(W, dtype=np.int64)
        self._check(self.lib.mirp_exchange_bytes(self.h, blob if blob else None, cnt.ctypes.data, C.byref(recv), rcnt.ctypes.data), "mirp_exchange_bytes")
        tot = int(rcnt.sum())
        raw = C.string_at(recv.value, tot) if tot else b""
        self.lib.mirp_free(recv)
        out, o = [], 0
        for r in range(W):
            out.append(raw[o:o + int(rcnt[r])]); o += int(rcnt[r])
        return out

    def set_fold_split_path(self, mode):
        """0: multiloop splits over split candidates (default), 1: the dense split loop for every window (same tables either way)."""
        self._check(self.lib.mirp_set_fold_split_path(self.h, int(mode)), "mirp_set_fold_split_path")

    def last_fold_dense(self):
        return int(self.lib.mirp_last_fold_dense(self.h))

    def set_coverage_path(self, mode):
        """-1: by record density (default), 0: atomic scatter, 1: fused scan where the input allows it."""
        self._check(self.lib.mirp_set_coverage_path(self.h, int(mode)), "mirp_set_coverage_path")

    def last_coverage_fused(self):
        """True when the last coverage pass built its tiles from the sorted records in LDS (dense inputs), False for the atomic scatter path."""
        return bool(self.lib.mirp_last_coverage_fused(self.h))

    def fold_summary(self):
        vp = C.c_void_p
        nl, mfe, st, n = vp(), vp(), vp(), C.c_int64()
        self._check(self.lib.mirp_get_fold_summary(self.h, C.byref(nl), C.byref(mfe), C.byref(st), C.byref(n)), "mirp_get_fold_summary")
        return {"n_lines": _copy_out(self.lib, nl, np.int32, n.value), "mfe": _copy_out(self.lib, mfe, np.int32, n.value),
                "status": _copy_out(self.lib, st, np.int32, n.value)}

    def fold_status(self):
        return self.fold_summary()["status"]

    def write_fold_text(self, fasta_path, out_path, wait=True):
        """RNALfold-format text of the resident fold output.  wait=False: returns once the output is off the device; formatting and writing go on
        in the library's worker threads (wait_text joins them)."""
        fn = self.lib.mirp_write_fold_text if wait else self.lib.mirp_write_fold_text_async
        self._check(fn(self.h, str(fasta_path).encode(), str(out_path).encode()), "mirp_write_fold_text")

    def wait_text(self):
        self._check(self.lib.mirp_wait_text(self.h), "mirp_wait_text")

    def write_depth_text(self, path, contig_names):
        """bam.depth.cut<CUT> of the last candidate() (MP:937-949), written by the library's worker threads."""
        blob = b"".join(n.encode() + b"\0" for n in contig_names)
        self._check(self.lib.mirp_write_depth_text(self.h, str(path).encode(), blob, len(contig_names)), "mirp_write_depth_text")

    def write_window_fasta(self, path, contig_names):
        """<prefix>.rnalfold.in_<i>.fa of the last candidate(): header + sequence of every window (MP:1124-1142)."""
        blob = b"".join(n.encode() + b"\0" for n in contig_names)
        self._check(self.lib.mirp_write_window_fasta(self.h, str(path).encode(), blob, len(contig_names)), "mirp_write_window_fasta")

    def get_fold(self):
        vp = C.c_void_p
        lines, ss, nl, mfe, st = vp(), vp(), vp(), vp(), vp()
        stride, ml = C.c_int32(), C.c_int32()
        self._check(self.lib.mirp_get_fold(self.h, C.byref(lines), C.byref(ss), C.byref(stride), C.byref(ml), C.byref(nl), C.byref(mfe), C.byref(st)),
                    "mirp_get_fold")
        n, stride, ml = self._n_windows, stride.value, ml.value
        raw = {"lines": _copy_out(self.lib, lines, FOLD_LINE_DTYPE, n * ml).reshape(n, ml),
               "ss": _copy_out(self.lib, ss, np.uint8, n * ml * stride).reshape(n, ml, stride), "stride": stride, "max_lines": ml,
               "n_lines": _copy_out(self.lib, nl, np.int32, n), "mfe": _copy_out(self.lib, mfe, np.int32, n),
               "status": _copy_out(self.lib, st, np.int32, n)}
        raw["overflow"] = self.fold_overflow()
        return raw

    def fold_overflow(self):
        """Windows of the last fold() that needed more structure lines than the main buffers hold (mirp_get_fold_overflow):
        {window index: (lines[max_lines2], ss[max_lines2, stride])}; use fold_window_lines() to read any window's lines."""
        vp = C.c_void_p
        wl, lines, ss, nl = vp(), vp(), vp(), vp()
        n, stride, ml = C.c_int64(), C.c_int32(), C.c_int32()
        self._check(self.lib.mirp_get_fold_overflow(self.h, C.byref(wl), C.byref(n), C.byref(lines), C.byref(ss), C.byref(stride), C.byref(ml), C.byref(nl)),
                    "mirp_get_fold_overflow")
        n, stride, ml = n.value, stride.value, ml.value
        wins = _copy_out(self.lib, wl, np.int32, n)
        a_l = _copy_out(self.lib, lines, FOLD_LINE_DTYPE, n * ml).reshape(n, ml)
        a_s = _copy_out(self.lib, ss, np.uint8, n * ml * stride).reshape(n, ml, stride)
        _copy_out(self.lib, nl, np.int32, n)
        return {int(w): (a_l[k], a_s[k]) for k, w in enumerate(wins)}

    def predict_raw(self, n_samples, min_mature_len, max_mature_len, allow_3nt, allow_no_star, minlen=55):
        """mirp_predict with the result left in flat arrays: {"result": MIRNA_DTYPE[n], "text": uint8[n, stride] structure rows, "n_passed", "status"};
        nothing is turned into Python objects per locus (write_result_reports takes the arrays as they are)."""
        from . import records
        pp = (C.c_int32 * 6)(int(n_samples), int(min_mature_len), int(max_mature_len), 1 if allow_3nt else 0, 1 if allow_no_star else 0, int(minlen))
        vp = C.c_void_p
        res, text, npass, stat = vp(), vp(), vp(), vp()
        nres, nw, stride = C.c_int64(), C.c_int64(), C.c_int32()
        self._check(self.lib.mirp_predict(self.h, pp, C.byref(res), C.byref(nres), C.byref(text), C.byref(stride), C.byref(npass), C.byref(stat), C.byref(nw)),
                    "mirp_predict")
        r = _copy_out(self.lib, res, records.MIRNA_DTYPE, nres.value)
        t = _copy_out(self.lib, text, np.uint8, nres.value * stride.value).reshape(nres.value, max(stride.value, 1))
        return {"result": r, "text": t, "n_passed": _copy_out(self.lib, npass, np.int32, nw.value), "status": _copy_out(self.lib, stat, np.int32, nw.value)}

    def fold_predict_report_stream(self, span, params, n_chunks, contig_names, contig_arrays, alns, sample_names, mirbase_form, outdir, prefix, max_lines=96):
        """mirp_fold_predict_report_stream: fold + filter of the resident windows in chunks, a chunk's read-mapping files written by a host thread while the
        device folds the next chunk, the seven report files at the end.  params = (n_samples, min_mature_len, max_mature_len, allow_3nt, allow_no_star,
        minlen).  -> (number of miRNA loci, chunks used, device seconds {fold, predict})."""
        pp = (C.c_int32 * 6)(*[int(x) for x in params])
        alns = np.ascontiguousarray(alns)
        keep = [np.ascontiguousarray(a, dtype=np.uint8) if a is not None and len(a) else None for a in contig_arrays]
        ptrs = (C.c_void_p * max(len(keep), 1))(*[a.ctypes.data if a is not None else None for a in keep])
        lens = np.array([len(a) if a is not None else 0 for a in keep] or [0], dtype=np.int64)
        blob = lambda xs: b"".join(x.encode() + b"\0" for x in xs)
        nl, nc, ms = C.c_int64(), C.c_int32(), (C.c_double * 2)()
        fn = self.lib.mirp_fold_predict_report_stream
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_int32,
                       C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        self._check(fn(self.h, int(span), int(max_lines), pp, int(n_chunks), blob(contig_names), len(contig_names), ptrs, lens.ctypes.data,
                       alns.ctypes.data if len(alns) else None, len(alns), blob(sample_names), len(sample_names), blob(mirbase_form), str(outdir).encode(), str(prefix).encode(),
                       C.byref(nl), C.byref(nc), ms), "mirp_fold_predict_report_stream")
        return int(nl.value), int(nc.value), {"fold_s": ms[0] / 1e3, "predict_s": ms[1] / 1e3}

    def select_windows(self, first, count):
        """mirp_select_windows: the fold and the filter run on windows [first, first + count) until changed; count < 0 = the whole list again."""
        self.lib.mirp_select_windows.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        self.lib.mirp_select_windows.restype = C.c_int
        self._check(self.lib.mirp_select_windows(self.h, int(first), int(count)), "mirp_select_windows")

    def predict(self, n_samples, min_mature_len, max_mature_len, allow_3nt, allow_no_star, minlen=55):
        from . import records
        pp = (C.c_int32 * 6)(int(n_samples), int(min_mature_len), int(max_mature_len), 1 if allow_3nt else 0, 1 if allow_no_star else 0, int(minlen))
        vp = C.c_void_p
        res, text, npass, stat = vp(), vp(), vp(), vp()
        nres, nw, stride = C.c_int64(), C.c_int64(), C.c_int32()
        self._check(self.lib.mirp_predict(self.h, pp, C.byref(res), C.byref(nres), C.byref(text), C.byref(stride), C.byref(npass), C.byref(stat), C.byref(nw)),
                    "mirp_predict")
        r = _copy_out(self.lib, res, records.MIRNA_DTYPE, nres.value)
        t = _copy_out(self.lib, text, np.uint8, nres.value * stride.value).reshape(nres.value, stride.value)
        tb, st = t.tobytes(), int(stride.value)
        ss = [tb[o:o + l].decode("ascii") for o, l in zip(range(0, len(tb), st), r["ss_len"].tolist())] if len(r) else []
        return {"result": r, "ss": ss, "n_passed": _copy_out(self.lib, npass, np.int32, nw.value), "status": _copy_out(self.lib, stat, np.int32, nw.value)}

    def predict_reasons(self, n_samples, min_mature_len, max_mature_len, allow_3nt, allow_no_star, minlen=55):
        """-d mode: int32 records [n, stride] of mirp_predict_reasons (layout in include/mirprefer.h), sorted by (window, mature, structure);
        the per-window records (mature index -1) come first within their window."""
        pp = (C.c_int32 * 6)(int(n_samples), int(min_mature_len), int(max_mature_len), 1 if allow_3nt else 0, 1 if allow_no_star else 0, int(minlen))
        rec, n, stride = C.c_void_p(), C.c_int64(), C.c_int32()
        self._check(self.lib.mirp_predict_reasons(self.h, pp, C.byref(rec), C.byref(n), C.byref(stride)), "mirp_predict_reasons")
        a = _copy_out(self.lib, rec, np.int32, n.value * stride.value).reshape(n.value, stride.value)
        if len(a):
            a = a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
        return a

    def last_fold_kernel_ms(self):
        """(fill kernel ms, epilogue kernel ms) of the last fold(), HIP events on the context's stream."""
        ms = (C.c_double * 2)()
        self.lib.mirp_last_fold_kernel_ms(self.h, ms)
        return ms[0], ms[1]

    def microbench(self):
        """Measured roofs of the fill kernel on this GPU (mirp_microbench): wave-instructions per second."""
        out = (C.c_double * 4)()
        self._check(self.lib.mirp_microbench(self.h, out), "mirp_microbench")
        return {"ds_read_b32_per_s": out[0], "ds_read_u16_per_s": out[1], "valu_pk16_per_s": out[2], "valu_u32_per_s": out[3]}

    def last_timings(self):
        ms = (C.c_double * 4)()
        self.lib.mirp_last_timings(self.h, ms)
        return {"coverage_ms": ms[0], "candidate_rest_ms": ms[1], "fold_ms": ms[2], "predict_ms": ms[3]}
