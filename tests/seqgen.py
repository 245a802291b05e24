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


def stress_family(r, k):
    """The five sequence families of the differential stress: 0 mixed (window()), 1 short tandem repeats, 2 two- / three-letter alphabets (energy
    ties), 3 perfect / near-perfect long hairpins, 4 N-rich."""
    n = r.randint(40, 350)
    if k == 0:
        return window(r, 60, 350)
    if k == 1:      # short tandem repeats
        unit = "".join(r.choice("ACGU") for _ in range(r.randint(1, 7)))
        return (unit * (n // len(unit) + 1))[:n]
    if k == 2:      # two-letter alphabets
        ab = r.choice(["GC", "AU", "GU", "ACG", "AGU"])
        return "".join(r.choice(ab) for _ in range(n))
    if k == 3:      # perfect / near-perfect long hairpins
        arm = r.randint(20, 160)
        a = "".join(r.choice("ACGU") for _ in range(arm))
        rc = {"A": "U", "C": "G", "G": "C", "U": "A"}
        b = [rc[c] for c in reversed(a)]
        for _ in range(r.randint(0, 6)):
            b[r.randrange(arm)] = r.choice("ACGU")
        return (a + "".join(r.choice("ACGU") for _ in range(r.randint(3, 12))) + "".join(b))[:350]
    return "".join(r.choice("ACGUN") if r.random() < 0.3 else r.choice("ACGU") for _ in range(n))      # N-rich


def microsatellites():
    """Low-complexity windows whose vienna-1.8.5 pair pools outgrow what one in-place compaction holds (4,096 entries) long before they outgrow the
    room behind a short window's triangle: (AU)k, (AU)k GCGC (AU)k, (GU)k, (ACGU)k at window lengths 100 .. 300."""
    out = []
    for n in list(range(100, 301, 10)) + [150, 199, 201, 290]:
        out.append(("AU" * 200)[:n])
        half = (n - 4) // 2
        out.append(("AU" * 200)[:half] + "GCGC" + ("AU" * 200)[:half])
        out.append(("GU" * 200)[:n])
        out.append(("ACGU" * 100)[:n])
        out.append(("AAUU" * 100)[:n])
    return out


def long_windows():
    """Windows beyond the LDS-resident fold kernels' reach (360 .. 480 nt; folded at spans 400 and 330 in tests/golden/long_folds.json.gz): 160 mixed ones
    and 40 of the stress families stretched to that length."""
    seqs = windows(4001, 160, 360, 480)
    r = random.Random(4002)
    for i in range(40):
        s = stress_family(r, i % 5)
        seqs.append((s + "".join(r.choice("ACGU") for _ in range(r.randint(20, 90))) + s[::-1] + s)[:r.randint(360, 470)])
    return seqs


def xl_windows():
    """PRECURSOR_LEN at the reference's upper limit (3000, MP:167-184): a window of 3,020 nt and one of 1,500 nt, folded at span 3000 in
    tests/golden/xl_folds.json.gz."""
    r = random.Random(3000)
    s = "".join(r.choice("ACGU") for _ in range(3020))
    return [s, s[700:2200]]
