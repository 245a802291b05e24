"""Host stage drivers mirroring the reference's run_prepare / run_candidate / run_fold / run_predict
(/root/reference/miR_PREFeR.py:3320-3627) on top of the C-ABI (capi.Context).  The stage checkpoint file
`<tmp>/<prefix>_recover` keeps the reference's structure {last_stage, finished_stages{stage:{name:path}}, files{stage:[paths]}}
(MP:3337-3353, 3412-3428, 3483-3490, 3607-3618) and a stage may start only if the previous one is recorded and its files exist
(previous_stage_saved, MP:3129-3137).  Stage artefacts keep the reference's names and text formats where the reference's
own stages exchange text (depth file, FASTA, RNALfold output, gff3); binary artefacts are .npz instead of pickles/BAMs."""
import os
import pickle
import sys
import time

import numpy as np

from . import capi, ingest, records

STAGES = ["prepare", "candidate", "fold", "predict"]


def _msg(s):
    sys.stdout.write("%s    %s\n" % (time.strftime("%a, %d %b %Y %H:%M:%S", time.localtime()), s))
    sys.stdout.flush()


def load_recover_file(name):
    if not os.path.exists(name):
        return None
    with open(name, "rb") as f:
        return pickle.load(f)


def _save_recover(name, d):
    tmp = name + ".temp"
    with open(tmp, "wb") as f:
        pickle.dump(d, f)
        f.flush()
        os.fsync(f.fileno())
    os.rename(tmp, name)  # write-temp + fsync + rename, as the reference does for its fold checkpoints (MP:3066-3079)


def previous_stage_saved(recovername, stage):
    d = load_recover_file(recovername)
    if not d or stage not in d["finished_stages"]:
        return False
    return all(os.path.exists(p) for p in d["files"].get(stage, []))


def detect_stage_last_finished(recovername):
    d = load_recover_file(recovername)
    if not d:
        return None
    last = None
    for s in STAGES:
        if s in d["finished_stages"] and all(os.path.exists(p) for p in d["files"].get(s, [])):
            last = s
        else:
            break
    return last


class Pipeline:
    """One device context + the option dict; the stage methods can run in one process (pipeline verb) or one per process
    (stage verbs): a stage that finds no device-resident state re-creates it from the previous stages' artefacts."""

    def __init__(self, dict_option, device=0, fold_model="vienna-2.1.2"):
        self.opt = dict_option
        self.tmp = dict_option["TMPFOLDER"] or os.path.join(dict_option["OUTFOLDER"], dict_option["NAME_PREFIX"] + "_tmp")
        os.makedirs(dict_option["OUTFOLDER"], exist_ok=True)
        os.makedirs(self.tmp, exist_ok=True)
        self.recovername = os.path.join(self.tmp, dict_option["NAME_PREFIX"] + "_recover")
        self.ctx = capi.Context(device)
        self.ctx.set_fold_model(fold_model)
        self.state = None  # None / "candidate" / "fold"
        self.data = None

    def _p(self, name):
        return os.path.join(self.tmp, name)

    def _fail_stage(self):
        _msg("Error: can not start the pipeline from this stage, the files needed are not generated or have been removed/moved. "
             "Please run previous stages first, or run the pipeline in the recover mode to automatically continue from where the job was ceased.")
        sys.exit(-1)

    # ---- prepare (MP:3320-3358): SAM/FASTA ingest replaces sam2bam / cat / sort / expand / strand split
    def run_prepare(self):
        _msg("Starting preparing data for the 'candidate' stage.")
        names, lens, samples, alns = ingest.read_sams(self.opt["ALIGNMENT_FILE"])
        prepared = self._p("prepared.npz")
        np.savez(prepared, contig_names=np.array(names, dtype=object), contig_lens=lens, sample_names=np.array(samples, dtype=object), alns=alns,
                 allow_pickle=True)
        d = {"last_stage": "prepare", "finished_stages": {"prepare": {"preparedname": prepared}}, "files": {"prepare": [prepared]}}
        _save_recover(self.recovername, d)
        _msg("Done (prepare stage)\n")

    def _load_inputs(self):
        if self.data is not None:
            return
        d = load_recover_file(self.recovername)
        z = np.load(d["finished_stages"]["prepare"]["preparedname"], allow_pickle=True)
        names = [str(x) for x in z["contig_names"]]
        fa = dict(ingest.read_fasta(self.opt["FASTA_FILE"]))
        missing = [n for n in names if n not in fa]
        if missing:
            sys.stderr.write("Error: sequence %s of the SAM header is not in the FASTA file\n" % missing[0])
            sys.exit(-1)
        self.data = {"names": names, "lens": z["contig_lens"], "samples": [str(x) for x in z["sample_names"]], "alns": z["alns"],
                     "contigs": [(n, fa[n]) for n in names]}
        self.ctx.load_genome(self.data["contigs"])
        self.ctx.load_alignments(self.data["alns"])

    def _ensure_candidate(self):
        if self.state in ("candidate", "fold"):
            return
        self._load_inputs()
        order = np.argsort(np.array(self.data["names"], dtype=object), kind="stable").astype(np.int32)  # sorted(dict_contigs), MP:1309
        self.counts = self.ctx.candidate(self.opt["READS_DEPTH_CUTOFF"], self.opt["MAX_GAP"], self.opt["PRECURSOR_LEN"], order)
        self.state = "candidate"

    # ---- candidate (MP:3361-3438)
    def run_candidate(self):
        if not previous_stage_saved(self.recovername, "prepare"):
            self._fail_stage()
        _msg("Starting identifying candidate regions")
        self.state = None
        self._ensure_candidate()
        names, prefix = self.data["names"], self.opt["NAME_PREFIX"]
        depthname = self._p("bam.depth.cut%d" % self.opt["READS_DEPTH_CUTOFF"])
        with open(depthname, "w") as f:
            f.write(records.depth_text(self.ctx.get_depth(), names))
        loci, psorted = self.ctx.get_loci()
        lociname = self._p(prefix + "_loci_dump.dump")
        with open(lociname, "wb") as f:
            pickle.dump(records.loci_to_dict(loci, psorted, names, self.opt["PRECURSOR_LEN"]), f, protocol=2)
        w = self.ctx.get_windows()
        fastaname = self._p(prefix + ".rnalfold.in_0.fa")
        with open(fastaname, "w") as f:
            for win in w["windows"]:
                f.write(records.fasta_header(win, w["wpeaks"], w["matures"], names) + "\n")
                f.write(w["seq"][win["seq_off"]:win["seq_off"] + win["seq_len"]].tobytes().decode() + "\n")
        dumpname = self._p(prefix + ".alndump_0.npz")
        np.savez(dumpname, windows=w["windows"], wpeaks=w["wpeaks"], matures=w["matures"])
        d = load_recover_file(self.recovername)
        d["last_stage"] = "candidate"
        d["finished_stages"]["candidate"] = {"depthfilename": depthname, "loci_dump_name": lociname, "fasta": [fastaname], "infodump": [dumpname],
                                             "num_loci": int(self.counts[1]), "num_fasta": int(self.counts[2])}
        d["files"]["candidate"] = [fastaname, dumpname]
        _save_recover(self.recovername, d)
        sys.stdout.write("%d candidate loci generated, %d regions to fold.\n" % (self.counts[1], self.counts[2]))
        _msg("Done (candidate stage)\n")

    def _fold_device(self):
        """Fold every window on the device; returns the per-window status array.  A window can produce more structure lines than the
        default capacity of 96 (tandem repeats do: one line per start position is possible); RNALfold has no such limit, so the stage is
        repeated once with the capacity no window can exceed."""
        self.ctx.fold(self.opt["PRECURSOR_LEN"])
        status = self.ctx.fold_status()
        if np.any(status == 1):
            self.ctx.fold(self.opt["PRECURSOR_LEN"], max_lines=self.opt["PRECURSOR_LEN"] + 52)
            status = self.ctx.fold_status()
        return status

    # ---- fold (MP:3441-3495)
    def run_fold(self, write_text=True):
        if not previous_stage_saved(self.recovername, "candidate"):
            self._fail_stage()
        _msg("Starting folding candidate sequences.")
        self._ensure_candidate()
        status = self._fold_device()
        self.state = "fold"
        prefix = self.opt["NAME_PREFIX"]
        foldname = self._p(prefix + "_rnalfoldoutput_0")
        bad = np.nonzero(status != 0)[0]
        if len(bad):
            sys.stderr.write("Error occurred when folding sequences (window %d, status %d).\n" % (bad[0], status[bad[0]]))
            sys.exit(-1)
        if write_text:
            d = load_recover_file(self.recovername)
            self.ctx.write_fold_text(d["finished_stages"]["candidate"]["fasta"][0], foldname)
        else:
            open(foldname, "w").close()
        d = load_recover_file(self.recovername)
        d["last_stage"] = "fold"
        d["finished_stages"]["fold"] = {"foldnames": [foldname]}
        d["files"]["fold"] = [foldname]
        _save_recover(self.recovername, d)
        _msg("Done (fold stage)\n")

    # ---- predict (MP:3498-3627); only the loci list, gff3 and fasta/ss outputs (report writers are out of scope)
    def run_predict(self):
        if not previous_stage_saved(self.recovername, "fold"):
            self._fail_stage()
        _msg("Starting predicting miRNAs.")
        if self.state != "fold":
            self._ensure_candidate()
            self._fold_device()
            self.state = "fold"
        out = self.ctx.predict(len(self.data["samples"]), self.opt["MIN_MATURE_LEN"], self.opt["MAX_MATURE_LEN"], self.opt["ALLOW_3NT_OVERHANG"],
                               self.opt["ALLOW_NO_STAR_EXPRESSION"])
        result = result_records(out, self.data["names"])
        prefix, outdir = self.opt["NAME_PREFIX"], self.opt["OUTFOLDER"]
        if not result:
            _msg("0 miRNA identified. No result files generated.")
            return result
        adjust_mature_star(result)
        gffname = os.path.join(outdir, prefix + "_miRNA.gff3")
        write_gff(result, gffname)
        maturename = os.path.join(outdir, prefix + "_miRNA.mature.fa")
        stemloopname = os.path.join(outdir, prefix + "_miRNA.precursor.fa")
        ssname = os.path.join(outdir, prefix + "_miRNA.precursor.ss")
        write_fasta_ss(result, dict(self.data["contigs"]), maturename, stemloopname, ssname)
        with open(self._p(prefix + "_miRNA.info.dump"), "wb") as f:
            pickle.dump(result, f)
        d = load_recover_file(self.recovername)
        d["last_stage"] = "predict"
        d["finished_stages"]["predict"] = {"gffname": gffname, "maturename": maturename, "stemloopname": stemloopname}
        d["files"]["predict"] = [gffname, maturename, stemloopname]
        _save_recover(self.recovername, d)
        _msg("The output files are in " + outdir)
        sys.stdout.write("%d miRNAs identified.\n" % len(result))
        _msg("Done (predict stage)\n")
        return result

    def run_pipeline(self):
        self.run_prepare()
        self.run_candidate()
        self.run_fold()
        return self.run_predict()

    def run_recover(self):
        last = detect_stage_last_finished(self.recovername)
        nxt = STAGES[STAGES.index(last) + 1:] if last else STAGES
        if last:
            _msg("Last finished stage: %s. Continue from the next stage." % last)
        for s in nxt:
            getattr(self, "run_" + s)()


def result_records(out, names):
    """`result` list of gen_miRNA_loci_nopredict (MP:2492-2496): [chr, fold_s, fold_e, mat_s, mat_e, star_s, star_e, ss, strand, has_star, exprinfo]."""
    res = []
    for m, ss in zip(out["result"], out["ss"]):
        f = int(m["reserved"])
        info = {"total_depth_mature": int(m["total_depth_mature"]), "total_depth_star": int(m["total_depth_star"])}
        if f & 1:
            info["max_imperfect_star"] = 1 if f & 8 else 0   # presence and zero/non-zero are what the gff writer reads (MP:2631)
            info["imperfect_star_which"] = ((f >> 1) & 3) - 1
        res.append([names[m["tid"]], int(m["fold_s"]), int(m["fold_e"]), int(m["mat_s"]), int(m["mat_e"]), int(m["star_s"]), int(m["star_e"]), ss,
                    records.STRAND[m["strand"]], bool(m["has_star"]), info])
    return res


def adjust_mature_star(resultlist):
    """MP:2611-2617: the more abundant arm is reported as the mature."""
    for m in resultlist:
        e = m[-1]
        if e["total_depth_mature"] < e["total_depth_star"]:
            e["total_depth_mature"], e["total_depth_star"] = e["total_depth_star"], e["total_depth_mature"]
            m[3], m[5] = m[5], m[3]
            m[4], m[6] = m[6], m[4]
            e["switch"] = True


def write_gff(resultlist, gffname):
    """gen_gff_from_result (MP:2619-2641), write_gff_line (MP:219-229)."""
    resultlist.sort(key=lambda m: m[:10])
    with open(gffname, "w") as f:
        for idx, m in enumerate(resultlist):
            e = m[-1]
            star = "n" if e["total_depth_star"] == 0 else "y"
            overhang = "2:2"
            if "max_imperfect_star" in e and e["max_imperfect_star"] != 0:
                overhang = "3:3" if e["imperfect_star_which"] == 2 else "2:3"
            other = "mature_expressed=y;star_expressed=" + star + ";overhangsize=" + overhang
            pre, mat = "miRNA-precursor_%d" % idx, "miRNA_%d" % idx
            f.write("\t".join([m[0], "miR-PREFeR", "miRNA-precursor", str(m[1]), str(m[2] - 1), ".", m[8], ".", "ID=%s;NAME=%s;Other=%s" % (pre, pre, other)]) + "\n")
            f.write("\t".join([m[0], "miR-PREFeR", "miRNA", str(m[3]), str(m[4] - 1), ".", m[8], ".", "ID=%s;NAME=%s;Other=" % (mat, mat)]) + "\n")


_RC = bytes.maketrans(b"ATGCU", b"UACGA")


def _region_seq(contigs, chrom, s, e, strand):
    seq = contigs[chrom][s - 1:e - 1].tobytes()
    return seq.translate(_RC)[::-1].decode() if strand == "-" else seq.decode()


def write_fasta_ss(resultlist, contigs, maturename, stemloopname, ssname):
    """Mature / precursor FASTA and structure file in the spirit of gen_mirna_fasta_ss_from_result (MP:2963-3019)."""
    with open(maturename, "w") as fm, open(stemloopname, "w") as fp, open(ssname, "w") as fs:
        for idx, m in enumerate(resultlist):
            pre = _region_seq(contigs, m[0], m[1], m[2], m[8])
            mat = _region_seq(contigs, m[0], m[3], m[4], m[8])
            fm.write(">miRNA_%d %s:%d-%d %s\n%s\n" % (idx, m[0], m[3], m[4] - 1, m[8], mat))
            fp.write(">miRNA-precursor_%d %s:%d-%d %s\n%s\n" % (idx, m[0], m[1], m[2] - 1, m[8], pre))
            fs.write(">miRNA-precursor_%d %s:%d-%d %s\n%s\n%s\n" % (idx, m[0], m[1], m[2] - 1, m[8], pre, m[7]))
