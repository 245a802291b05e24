#!/usr/bin/env python3
"""Dev-only (build container): golden vectors for the structure parsing / duplex rules of the predict stage,
from DIRECT calls of the reference's own functions through the py3 shim (ref_shim.py; nothing of the reference is copied):

  * `structures`: seeded dot-bracket strings (stem-loops with bulges and interior loops, multi-branch structures, parallel stems, dangling
    ends) -> is_stem_loop(ss, 3), has_one_good_bifurcation(ss), filter_ss(ss) and the list get_structures_next_extendregion yields for an
    RNALfold-output file holding that one line (MP:1541-1724);
  * `maturestar`: (ss, mature, foldstart, region, strand) -> get_maturestar_info (MP:1876-1999) -- the fail string, or the 9-tuple, or the
    exception class the reference raises -- with stat_duplex / pass_stat_duplex (MP:1815-1873) of the same duplex when it gets that far.
    The generator steers duplex shapes so that EVERY failure code occurs at least 20 times, including FAIL_STRUCTURE_TOO_MANY_BULGE_OR_LOOP and
    FAIL_STRUCTURE_MATCHED_BASES, which no pipeline fixture reaches.

Output: tests/golden/struct_rules.json.gz"""
import gzip, json, os, random, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
GOLD = os.path.dirname(HERE)


def stem(r, feats, loop):
    """A stem-loop from the outside in: feats = [(pairs, unpaired5, unpaired3), ...] -- `pairs` stacked pairs, then an interior loop / bulge."""
    left, right = "", ""
    for k, a, b in feats:
        left += "(" * k + "." * a
        right = "." * b + ")" * k + right
    return left + "." * loop + right


def random_feats(r, kind):
    n = r.randint(2, 9)
    feats = []
    for i in range(n):
        k = r.randint(2, 9)
        if kind == "clean":
            a = b = 0 if r.random() < 0.7 else r.choice([1, 1, 2])
        elif kind == "bulgy":
            a, b = r.choice([(0, 1), (1, 0), (2, 0), (0, 2), (1, 2), (3, 0), (0, 3), (1, 1), (2, 2)])
        elif kind == "loopy":
            a = b = r.choice([0, 1, 1, 2, 2, 3, 4])
        elif kind == "many":          # many small features inside a ~22-nt duplex: more than 5 bulges / loops
            k = r.randint(1, 2)
            a, b = r.choice([(1, 1), (1, 0), (0, 1), (1, 1), (2, 2)])
        else:
            a, b = r.randint(0, 3), r.randint(0, 3)
        feats.append((k, a, b))
    k, a, b = feats[-1]
    feats[-1] = (k, 0, 0)
    return feats


def random_structure(r):
    kind = r.choice(["clean", "bulgy", "loopy", "many", "mixed", "multi", "parallel", "bif"])
    if kind in ("clean", "bulgy", "loopy", "many", "mixed"):
        s = stem(r, random_feats(r, kind), r.randint(3, 12))
    elif kind == "parallel":
        s = "".join(stem(r, random_feats(r, "clean"), r.randint(3, 8)) + "." * r.randint(0, 6) for _ in range(r.randint(2, 3)))
    elif kind == "bif":           # one bifurcation inside an outer stem
        inner = stem(r, random_feats(r, "clean")[:3], r.randint(3, 7)) + "." * r.randint(0, 4) + stem(r, random_feats(r, "clean")[:3], r.randint(3, 7))
        k = r.randint(3, 20)
        s = "(" * k + "." * r.randint(0, 3) + inner + "." * r.randint(0, 3) + ")" * k
    else:                         # multi-branch
        inner = "".join(stem(r, random_feats(r, r.choice(["clean", "bulgy"]))[:r.randint(1, 4)], r.randint(3, 9)) + "." * r.randint(0, 5) for _ in range(r.randint(2, 4)))
        k = r.randint(1, 12)
        s = "(" * k + inner + ")" * k
    s = "." * r.choice([0, 0, 1, 2, 5]) + s + "." * r.choice([0, 0, 1, 3, 6])
    return s if len(s) <= 340 else random_structure(r)          # RNALfold lines are balanced: never truncate


def call(fn, *a):
    try:
        return {"ok": fn(*a)}
    except Exception as e:          # the reference raises on some inputs (KeyError, IndexError ...): recorded as such
        return {"raises": type(e).__name__}


def main():
    import ref_shim
    g = ref_shim.load_reference()
    r = random.Random(20261003)
    structures, seen = [], set()
    with tempfile.TemporaryDirectory() as tmp:
        while len(structures) < 4000:
            ss = random_structure(r)
            if len(ss) < 20 or ss in seen:
                continue
            seen.add(ss)
            energy = -r.randint(0, 9000) / 100.0
            start = r.randint(1, 40)
            fn = os.path.join(tmp, "o")
            with open(fn, "w") as f:
                f.write(">chr1:100-400 + 150-300 0 x M:1-2/+/3\n%s (%6.2f) %4d\nACGU\n (%6.2f)\n" % (ss, energy, start, energy))
            try:
                ext = list(g["get_structures_next_extendregion"](fn, 55, 3))
            except Exception:
                continue          # the reference itself raises on this string (it never sees such a line from RNALfold)
            assert len(ext) == 1
            structures.append({"ss": ss, "energy": energy, "start": start,
                               "is_stem_loop": bool(g["is_stem_loop"](ss, 3)),
                               "good_bifurcation": call(g["has_one_good_bifurcation"], ss),
                               "filter_ss": call(g["filter_ss"], ss),
                               "extend": [[float(e), int(s), x, int(t)] for e, s, x, t in ext[0][2]]})
    # ---- get_maturestar_info
    ms, counts = [], {}
    tmpdir = tempfile.mkdtemp()
    want_each = 25
    codes = ["FAIL_STRUCTURE_MATCHED_BASES", "FAIL_STRUCTURE_MATURE_NOT_IN_FOLD_REGION", "FAIL_STRUCTURE_MATURE_NOT_IN_ONE_ARM",
             "FAIL_STRUCTURE_MATURE_MATCH_SMALL_THAN_14", "FAIL_STRUCTURE_MATURE_STAR_OVERLAP", "FAIL_STRUCTURE_STAR_OUT_OF_FOLD_REGION",
             "FAIL_STRUCTURE_STAR_NOT_IN_ONE_ARM", "FAIL_STRUCTURE_TOO_MANY_BULGE_OR_LOOP", "FAIL_STRUCTURE_MAX_BULGE_LARGE_THAN_2",
             "FAIL_STRUCTURE_TOTAL_LOOP_SIZE_LARGER_THAN_5", "FAIL_STRUCTURE_NUM_BULGE_MORE_THAN_2", "OK", "RAISES"]
    tries = 0
    while (len(ms) < 6000 or any(counts.get(c, 0) < want_each for c in codes[:-1])) and tries < 400000:
        tries += 1
        kind = r.choice(["clean", "bulgy", "loopy", "many", "many", "mixed", "multi", "unbalanced", "bif"])
        if kind == "unbalanced":
            ss = random_structure(r)
            cut = r.randint(0, max(0, len(ss) // 3))
            ss = ss[:cut].replace("(", ".") + ss[cut:]          # opening brackets removed: a ')' meets an empty stack
        elif kind == "multi":
            ss = random_structure(r)
        elif kind == "bif":          # one bifurcation inside a long outer stem: a structure of type 1, where the star can straddle both inner stems
            inner = stem(r, random_feats(r, "clean")[:2], r.randint(3, 6)) + "." * r.randint(0, 3) + stem(r, random_feats(r, "clean")[:2], r.randint(3, 6))
            k = r.randint(12, 30)
            ss = "(" * k + "." * r.randint(0, 2) + inner + "." * r.randint(0, 2) + ")" * k
        else:
            ss = stem(r, random_feats(r, kind), r.randint(3, 25))
            ss = "." * r.choice([0, 0, 2, 4]) + ss + "." * r.choice([0, 0, 2, 4])
        if len(ss) < 40:
            continue
        n = len(ss)
        strand = r.choice("+-")
        foldstart = r.randint(1, 30)
        regionstart = r.randint(1000, 5000)
        regionend = regionstart + foldstart - 1 + n + r.randint(0, 30)
        ml = r.randint(18, 24)
        if r.random() < 0.08:        # mature outside the fold region
            l0 = r.choice([-r.randint(1, 6), n - ml + r.randint(1, 6)])
        else:
            l0 = r.randint(0, max(0, n - ml))
        m = g["pos_local_2_genome"](l0, l0 + ml, strand, regionstart, regionend, foldstart, foldstart + n)
        res = call(g["get_maturestar_info"], ss, (m[0], m[1]), foldstart, foldstart + n, regionstart, regionend, strand)
        if "raises" in res:
            code = "RAISES"
        elif isinstance(res["ok"], str):
            code = res["ok"]
        else:
            code = "OK"
            res["ok"] = [int(res["ok"][0]), int(res["ok"][1]), int(res["ok"][2]), int(res["ok"][3]), res["ok"][4], bool(res["ok"][5]), res["ok"][6],
                         int(res["ok"][7]), int(res["ok"][8])]
        # keep the rare codes always, the common ones up to a quota
        quota = 1500 if code == "OK" else 600
        if counts.get(code, 0) >= quota:
            continue
        counts[code] = counts.get(code, 0) + 1
        # does the predict stage hand this very string to get_maturestar_info?  (a line that is one structure by itself: a stem-loop, or a
        # piece-less good bifurcation) -> its structure type, else None
        whole = None
        if len(ss) >= 55:
            fn = os.path.join(tmpdir, "o")
            with open(fn, "w") as f:
                f.write(">chr1:100-400 + 150-300 0 x M:1-2/+/3\n%s (%6.2f) %4d\nACGU\n (%6.2f)\n" % (ss, -10.0, foldstart, -10.0))
            try:
                ext = list(g["get_structures_next_extendregion"](fn, 55, 3))[0][2]
                if len(ext) == 1 and ext[0][2] == ss and ext[0][1] == foldstart:
                    whole = int(ext[0][3])
            except Exception:
                pass
        ms.append({"whole": whole, "ss": ss, "mature": [int(m[0]), int(m[1])], "foldstart": foldstart, "regionstart": regionstart, "regionend": regionend, "strand": strand,
                   "result": res})
    # direct stat_duplex / pass_stat_duplex on seeded duplex halves
    duplex = []
    for _ in range(2000):
        feats = random_feats(r, r.choice(["clean", "bulgy", "loopy", "many", "mixed"]))
        left, right = "", ""
        for k, a, b in feats:
            left += "(" * k + "." * a
            right = "." * b + ")" * k + right
        if r.random() < 0.5:
            left, right = right, left          # mature on the 3' arm: the function swaps the bracket roles
        st = call(g["stat_duplex"], left, right)
        ps = call(g["pass_stat_duplex"], *st["ok"]) if "ok" in st else None
        duplex.append({"mature": left, "star": right, "stat": st, "pass": ps})
    out = {"generator": "reference functions through tests/golden/tools/ref_shim.py: is_stem_loop, has_one_good_bifurcation, filter_ss, "
                        "get_structures_next_extendregion (MP:1541-1724), get_maturestar_info, stat_duplex, pass_stat_duplex (MP:1815-1999)",
           "structures": structures, "maturestar": ms, "duplex": duplex, "maturestar_code_counts": counts}
    path = os.path.join(GOLD, "struct_rules.json.gz")
    with gzip.open(path, "wt", compresslevel=9) as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes;", len(structures), "structures,", len(ms), "maturestar cases,", len(duplex), "duplexes")
    for c in codes:
        print("  %-48s %d" % (c, counts.get(c, 0)))


if __name__ == "__main__":
    main()
