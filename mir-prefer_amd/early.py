"""What the CLI starts BEFORE the heavy imports: the library load, the device context and the genome read, each on a thread of its own.

Opening the device is kernel time nobody can shorten from user space: on the bench's MI355X box `open("/dev/kfd")` alone takes 0.13 s, hipInit
0.16 - 0.25 s, the first stream another 0.02 - 0.15 s (profiles/r5_hip_startup.txt: profiles/tools/ctx_probe.cpp under the call timer sysprobe.c).
ctypes releases the interpreter lock during a native call, so mirp_create runs while the main thread imports numpy, parses the configuration and
tokenizes the SAM files, and mirp_read_fasta reads the genome at the same time.  This module imports nothing but ctypes (8 ms); capi.py adopts the
handles (capi.Context, capi.read_fasta).  Nothing here catches a failure: a missing library or GPU surfaces in capi exactly as without the early start."""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIRP_LIB") or os.path.join(_HERE, "libmirprefer.so")   # MIRP_LIB: dev tools load the diagnostics build (make DIAG=1)


class FastaData(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("names", C.c_void_p), ("len", C.POINTER(C.c_int64)), ("seq", C.c_void_p), ("n_bytes", C.c_int64)]


class SamData(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("contig_names", C.c_void_p), ("contig_len", C.POINTER(C.c_int64)), ("n_samples", C.c_int32),
                ("sample_names", C.c_void_p), ("alns", C.c_void_p), ("n_alns", C.c_int64), ("segs", C.c_void_p), ("n_segs", C.c_int64)]


_lib = None
_contexts = {}      # device -> (thread, handle, result box)
_fastas = {}        # path -> (thread, FastaData, error buffer, result box)
_ingests = {}       # tuple of SAM paths -> (thread, tokenizer handle, error buffer, result box)


def cdll():
    """The one CDLL object of the process (capi.load_library sets the prototypes on it); None when the library has not been built."""
    global _lib
    if _lib is None and os.path.exists(LIB_PATH):
        _lib = C.CDLL(LIB_PATH)
    return _lib


def start_context(device):
    lib = cdll()
    if lib is None or device in _contexts:
        return
    create = lib.mirp_create
    create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    create.restype = C.c_int
    h, box = C.c_void_p(), {}

    def run():
        box["rc"] = create(int(device), C.byref(h))
    th = threading.Thread(target=run, daemon=True)
    th.start()
    _contexts[device] = (th, h, box)


def take_context(device):
    """-> (return code of mirp_create, handle) of the context started for `device`, or None; a context is handed out once."""
    ent = _contexts.pop(device, None)
    if ent is None:
        return None
    th, h, box = ent
    th.join()
    return box.get("rc", -2), h


def start_fasta(path):
    lib = cdll()
    if lib is None or path in _fastas or str(path).endswith(".gz"):
        return
    fn = lib.mirp_read_fasta
    fn.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(FastaData), C.c_char_p, C.c_size_t]      # as capi.load_library declares it
    fn.restype = C.c_int
    d, err, box = FastaData(), C.create_string_buffer(512), {}

    def run():
        box["rc"] = fn(str(path).encode(), None, 0, C.byref(d), err, 512)
    th = threading.Thread(target=run, daemon=True)
    th.start()
    _fastas[path] = (th, d, err, box)


def take_fasta(path):
    """-> (return code of mirp_read_fasta, FastaData, error text) of the read started for `path`, or None; handed out once (the caller frees)."""
    ent = _fastas.pop(path, None)
    if ent is None:
        return None
    th, d, err, box = ent
    th.join()
    return box.get("rc", -1), d, err.value.decode()


def start_ingest(paths):
    """The host half of the SAM ingest (mirp_tokenize_sams: header + tokenizer threads) while the device is still being opened; the prepare stage hands the
    result to the device half (Context.ingest_tokenized: keep-region filter + stable radix sort) as soon as the context is there."""
    lib = cdll()
    key = tuple(str(p) for p in paths)
    if lib is None or not key or key in _ingests or any(p.endswith(".gz") for p in key):
        return
    fn = lib.mirp_tokenize_sams
    fn.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]      # as capi.load_library declares it
    fn.restype = C.c_int
    arr = (C.c_char_p * len(key))(*[p.encode() for p in key])
    tok, err, box = C.c_void_p(), C.create_string_buffer(512), {}

    def run():
        box["rc"] = fn(arr, len(key), 0, C.byref(tok), err, 512)
    th = threading.Thread(target=run, daemon=True)
    th.start()
    _ingests[key] = (th, tok, err, box)


def has_ingest(paths):
    return tuple(str(p) for p in paths) in _ingests


def take_ingest(paths):
    """-> (return code of mirp_tokenize_sams, handle, error text) of the tokenizer run started for these paths, or None; handed out once (the caller passes the
    handle to mirp_ingest_tokenized_gpu, which releases it)."""
    ent = _ingests.pop(tuple(str(p) for p in paths), None)
    if ent is None:
        return None
    th, d, err, box = ent
    th.join()
    return box.get("rc", -1), d, err.value.decode()
