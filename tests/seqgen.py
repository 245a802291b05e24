"""Seeded synthetic precursor windows for fold parity tests (random, planted hairpins, N runs, soft-masked)."""
import random

_RC = {"A": "U", "C": "G", "G": "C", "U": "A"}


def window(r, lo, hi):
    n = r.randint(lo, hi)
    gc = r.choice([0.3, 0.5, 0.65])
    mode = r.random()
    s = "".join(r.choice("GC") if r.random() < gc else r.choice("AU") for _ in range(n))
    if mode < 0.4 and n > 80:
        arm = r.randint(18, 34)
        loop = r.randint(3, 40)
        a = "".join(r.choice("ACGU") for _ in range(arm))
        b = [_RC[c] for c in reversed(a)]
        for _ in range(r.randint(0, 4)):
            b[r.randrange(arm)] = r.choice("ACGU")
        if r.random() < 0.3:
            b.insert(r.randrange(arm), r.choice("ACGU"))
        hp = a + "".join(r.choice("ACGU") for _ in range(loop)) + "".join(b)
        pos = r.randint(0, max(0, n - len(hp)))
        s = (s[:pos] + hp + s[pos + len(hp):])[:max(n, len(hp))]
    if mode > 0.9:
        p = r.randrange(len(s))
        s = s[:p] + "N" * r.randint(1, 5) + s[p:]
    if 0.8 < mode < 0.9:
        s = s.replace("U", "T").lower() if r.random() < 0.5 else s.replace("U", "t")
    return s


def windows(seed, count, lo, hi):
    r = random.Random(seed)
    return [window(r, lo, hi) for _ in range(count)]
