"""Window-level re-balancing of a contig-sharded run before the fold ("optional all-to-all of window payloads").

Whole contigs per rank (dist.partition_contigs) leave the ranks unevenly loaded -- 12 MSU7-sized contigs on 8 GPUs: 1.23 x the mean on the
fullest rank, 5 contigs occupy at most 5 ranks -- while the fold, 89 % of the step, is independent per window.  The reference itself fans
PIECES of the candidate list out to its workers, not contigs (/root/reference/miR_PREFeR.py:1329-1354 piece table, :2468-2480 one process
per piece).  After the candidate stage every rank knows all window counts; a rank above the mean ships the tail of its window list to ranks
below it (cut only between L/R pairs, MP:2394-2403), keeps the head (mirp_limit_windows) and the helpers fold + filter what they received
through the batch entry points of the C-ABI (mirp_fold_batch, mirp_predict_batch).  A payload carries everything those two need: window
records, peaks, candidate matures, sequences and the slice of the position-sorted alignment records that the windows can see.  The payloads
travel over the context's communicator (mirp_exchange_bytes: RCCL, or the local transport for ranks that share a GPU).  The loci of imported
windows join the helper's list; tids are genome-wide, so nothing downstream cares which rank found a locus."""
import io

import numpy as np

from . import records

import os

THRESHOLD = 1.1      # re-balance when the fullest rank holds more than this times the mean
MIN_MOVE = int(os.environ.get("MIRP_BALANCE_MIN_MOVE", "16"))      # windows; tests lower it to move windows of miniature inputs


def parse_units(tags):
    """Start index of every pairing unit of a window list -- a tag-0 window alone, otherwise two consecutive entries (filter_next_loci pairs L/R
    entries by position, MP:2394-2403) -- plus the end sentinel."""
    starts, k, n = [], 0, len(tags)
    t = np.asarray(tags).tolist()
    while k < n:
        starts.append(k)
        k += 1 if t[k] == 0 else 2
    starts.append(n)
    return np.array(starts, dtype=np.int64)


def plan(counts, threshold=THRESHOLD):
    """counts[r] = windows on rank r -> [(src, dst, n_windows)] in the order every donor cuts its tail; [] when the load is even enough."""
    counts = [int(x) for x in counts]
    W, T = len(counts), sum(counts)
    if W < 2 or T == 0 or max(counts) * W <= threshold * T:
        return []
    target = -(-T // W)
    deficit = [max(0, target - n) for n in counts]
    surplus = [max(0, n - target) for n in counts]
    small = MIN_MOVE      # a transfer below this is not worth a payload: the donor keeps those windows
    moves = []
    for s in sorted(range(W), key=lambda r: (-surplus[r], r)):
        for d in sorted(range(W), key=lambda r: (-deficit[r], r)):
            m = min(surplus[s], deficit[d])
            if m < small:
                continue
            moves.append((s, d, m))
            surplus[s] -= m
            deficit[d] -= m
    return moves


def _snap(units, b):
    """largest unit start <= b"""
    return int(units[np.searchsorted(units, b, side="right") - 1])


def export_ranges(moves, rank, n_windows, tags):
    """This rank's cuts: (n_keep, [(dst, a, b)]) with every boundary on a pairing unit."""
    mine = [(d, m) for s, d, m in moves if s == rank]
    if not mine:
        return n_windows, []
    units = parse_units(tags)
    keep = _snap(units, n_windows - sum(m for _, m in mine))
    out, a, acc = [], keep, n_windows - sum(m for _, m in mine)
    for k, (d, m) in enumerate(mine):
        acc += m
        b = n_windows if k == len(mine) - 1 else _snap(units, acc)
        if b > a:
            out.append((d, a, b))
        a = b
    return keep, out


def pack(win, alns, a, b, src):
    """Payload of windows [a, b) of a get_windows() result: self-contained arrays with offsets re-based to the payload.  The window list is laid
    out in order, so a contiguous range of windows owns contiguous ranges of the peak / mature / sequence arrays (gaps included)."""
    W = win["windows"][a:b].copy()

    def cut(arr, off, cnt):
        lo, hi = int(W[off].min()), int((W[off].astype(np.int64) + W[cnt]).max())
        W[off] -= lo
        return arr[lo:hi]
    pk = cut(win["wpeaks"], "peak_off", "n_peaks")
    mt = cut(win["matures"], "mature_off", "n_matures")
    sq = cut(win["seq"], "seq_off", "seq_len")
    # the records the windows can see: per contig, positions [min ws - 64, max we + 64] of the (tid, pos)-sorted array
    key = alns["tid"].astype(np.int64) << 32 | alns["pos"].astype(np.int64)
    parts = []
    for t in np.unique(W["tid"]):
        sel = W[W["tid"] == t]
        lo = np.searchsorted(key, (int(t) << 32) | max(int(sel["ws"].min()) - 64, 0), side="left")
        hi = np.searchsorted(key, (int(t) << 32) | (int(sel["we"].max()) + 64), side="right")
        parts.append(alns[lo:hi])
    buf = io.BytesIO()
    np.savez(buf, windows=W, wpeaks=pk, matures=mt, seq=sq, alns=np.concatenate(parts) if parts else alns[:0], meta=np.array([src, a, b], dtype=np.int64))
    return buf.getvalue()


def unpack(blob):
    z = np.load(io.BytesIO(blob))
    return {k: z[k] for k in z.files}


def exchange(xchg, rank, world, get_windows, alns, counts):
    """Collective.  xchg(blocks) -> blocks received by source rank (Context.exchange_bytes, or a host-object stand-in); get_windows() is only
    called on a rank that ships windows.  -> (n_keep, [imported payload dicts in source-rank order], moves); the caller limits its context
    to n_keep windows."""
    moves = plan(counts)
    if not moves:
        return int(counts[rank]), [], moves
    keep, blocks = int(counts[rank]), [b""] * world
    if any(s == rank for s, _, _ in moves):
        win = get_windows()
        keep, ranges = export_ranges(moves, rank, int(counts[rank]), win["windows"]["tag"])
        for dst, a, b in ranges:
            blocks[dst] = pack(win, alns, a, b, rank)
    got = xchg(blocks)
    return keep, [unpack(g) for g in got if len(g)], moves


def fold_imported(ctx, payload, span):
    """RNALfold -L on the windows of a payload (mirp_fold_batch); a payload that holds a window with more than 96 structure lines is folded
    again at the capacity no window exceeds, as mirp_fold does for such windows."""
    W, seq = payload["windows"], payload["seq"]
    seqs = [seq[w["seq_off"]:w["seq_off"] + w["seq_len"]].tobytes() for w in W]
    raw = ctx.fold_batch_raw(seqs, span)
    if (raw["status"] == 1).any():
        raw = ctx.fold_batch_raw(seqs, span, max_lines=int(W["seq_len"].max()) + 4)
    if (raw["status"] != 0).any():
        raise RuntimeError("fold of imported windows failed (status %d)" % int(raw["status"][raw["status"] != 0][0]))
    payload["fold_raw"] = raw
    return raw


def predict_imported(ctx, payload, params):
    """filter_next_loci over the windows of a payload (mirp_predict_batch + the L/R rule: R is looked at only if L failed, MP:2417-2431);
    -> {"result": MIRNA records, "ss": [text]} like Context.predict()."""
    W, raw = payload["windows"], payload["fold_raw"]
    mir, nm, st = ctx.predict_batch(W, payload["matures"], payload["alns"], raw, params)
    if (st != 0).any():
        raise RuntimeError("filter of imported windows failed (status %d)" % int(st[st != 0][0]))
    units = parse_units(W["tag"])
    second = np.ones(len(W), dtype=bool)
    second[units[:-1]] = False                                     # the R entry of a pair is looked at only if its L entry failed
    keep = (nm > 0) & ~(second & (np.concatenate([[0], nm[:-1]]) > 0))
    idx = np.nonzero(keep)[0]
    res = np.ascontiguousarray(mir[idx, 0]) if len(idx) else np.zeros(0, dtype=records.MIRNA_DTYPE)
    texts = [raw["ss"][k, int(m["line"]), int(m["ss_off"]):int(m["ss_off"]) + int(m["ss_len"])].tobytes().decode("ascii") for k, m in zip(idx.tolist(), res)]
    return {"result": res, "ss": texts}


def fold_text(payload, names):
    """RNALfold-format text of the imported windows (the helper's share of the fold stage's artefact, MP:3085-3098)."""
    W, raw, out = payload["windows"], payload["fold_raw"], []
    for k in range(len(W)):
        w = W[k]
        out.append(records.fasta_header(w, payload["wpeaks"], payload["matures"], names))
        for j in range(int(raw["n_lines"][k])):
            ln = raw["lines"][k, j]
            if ln["printed"]:
                out.append("%s (%6.2f) %4d" % (raw["ss"][k, j, :int(ln["len"])].tobytes().decode("ascii"), ln["energy"] / 100.0, ln["start"]))
        s = payload["seq"][w["seq_off"]:w["seq_off"] + w["seq_len"]].tobytes().decode("ascii").upper().replace("T", "U")
        out.append(s)
        out.append(" (%6.2f)" % (raw["mfe"][k] / 100.0))
    return "\n".join(out) + "\n" if out else ""
