"""Digests the headline / mid fixtures are stated in (tests/golden/tools/gen_headline_golden.py writes them, the tests recompute them)."""
import hashlib

import numpy as np


def fold_digest(lines, mfe):
    """6 bytes over the printed lines (structure, energy in 0.01 kcal/mol, start column) and the final MFE of one window."""
    h = hashlib.blake2b(digest_size=6)
    for ss, e, st in lines:
        h.update(("%s %d %d\n" % (ss, e, st)).encode())
    h.update(("|%d" % mfe).encode())
    return h.digest()


def seq_digest(s):
    return hashlib.blake2b(s.encode() if isinstance(s, str) else bytes(s), digest_size=6).digest()


def array_digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
