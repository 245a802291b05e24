"""Host stage drivers mirroring the reference's run_prepare / run_candidate / run_fold / run_predict
(/root/reference/miR_PREFeR.py:3320-3627) on top of the C-ABI (capi.Context).  The stage checkpoint file
`<tmp>/<prefix>_recover` keeps the reference's structure {last_stage, finished_stages{stage:{name:path}}, files{stage:[paths]}}
(MP:3337-3353, 3412-3428, 3483-3490, 3607-3618) and a stage may start only if the previous one is recorded and its files exist
(previous_stage_saved, MP:3129-3137).  Stage artefacts keep the reference's names and text formats where the reference's
own stages exchange text (depth file, FASTA, RNALfold output, gff3); binary artefacts are .npz instead of pickles/BAMs."""
import os
import pickle
import threading
import sys
import time

import numpy as np

from . import balance, capi, dist, early, gffmask, ingest, records

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

    _imported, _moves = (), None          # window-level re-balancing (balance.py): payloads received, the plan (None: not decided yet)
    lean, _prepared = False, None

    def __init__(self, dict_option, device=0, fold_model="vienna-2.1.2", rank=0, world=1, ctx=None, lean=False):
        """rank / world: contig sharding over one process per GPU (`torch.distributed` initialised by the caller, see cli.py).  Every rank
        runs the stages on its own contigs (dist.partition_contigs); rank 0 merges the artefacts and writes the result files.
        lean (the `pipeline` verb of ONE process without -k and without -d, cli.py): the run ends by deleting its temporary folder (run_removetmp,
        MP:3630-3639), so the stage artefacts nobody will read are not made at all -- prepared.npz, the depth file, the window FASTA, the loci dump,
        ExRegionA.gff3, the window dump, the RNALfold-format text (135 MB at config[1]) and the result pickle -- the stages hand their state on in
        memory, and the report files come from one native call.  No stage is recorded in `<prefix>_recover`: a lean run that dies leaves nothing to
        recover from and is simply run again (0.5 s); every other way of running -- stage verbs, `recover`, -k, -d, several ranks -- writes every
        file as before."""
        self.rank, self.world = rank, world
        self.lean = bool(lean) and world == 1 and not dict_option.get("OUTPUT_DETAILS_FOR_DEBUG")
        self._prepared = None
        self.opt = dict_option
        self.tmp = dict_option["TMPFOLDER"] or os.path.join(dict_option["OUTFOLDER"], dict_option["NAME_PREFIX"] + "_tmp")
        os.makedirs(dict_option["OUTFOLDER"], exist_ok=True)
        os.makedirs(self.tmp, exist_ok=True)
        self.recovername = os.path.join(self.tmp, dict_option["NAME_PREFIX"] + "_recover")
        self.ctx = ctx if ctx is not None else capi.Context(device)      # ctx: a context that already holds the ranks' RCCL communicator
        self.ctx.set_fold_model(fold_model)
        self.state = None  # None / "candidate" / "fold"
        self.data = None
        # the context's own RCCL communicator (cli.py / bench.py: mirp_dist_init) carries the data-path exchanges of a sharded run: the record routing
        # of the ingest and the gather of the loci list.  Without one (ranks that share a GPU in tests) those go through the host's object channel.
        self.rccl = world > 1 and self.ctx.dist_world() == world
        self._imported, self._moves = [], None          # window-level re-balancing (balance.py)

    def _p(self, name):
        return os.path.join(self.tmp, name)

    def _say(self, text):
        if self.rank == 0:
            _msg(text)

    def _barrier(self):
        if self.world > 1:
            import torch.distributed as tdist
            tdist.barrier()

    def _all_gather(self, obj):
        """Small python objects (file names, counts, the loci list) from every rank, in rank order."""
        if self.world == 1:
            return [obj]
        import torch.distributed as tdist
        out = [None] * self.world
        tdist.all_gather_object(out, obj)
        return out

    def _agree_ok(self, ok, what=""):
        """Every rank learns whether any rank failed BEFORE the next collective, so that all of them leave with exit status -1 together
        instead of the healthy ones hanging in a barrier until the watchdog fires (the reference's workers just die, MP:3103-3106)."""
        flags = self._all_gather(bool(ok))
        if not all(flags):
            if ok and self.rank == 0:
                sys.stderr.write("Error: rank %d failed%s; stopping all ranks.\n" % (flags.index(False), (" (" + what + ")") if what else ""))
            sys.exit(-1)

    def _fail_stage(self):
        self._say("Error: can not start the pipeline from this stage, the files needed are not generated or have been removed/moved. "
             "Please run previous stages first, or run the pipeline in the recover mode to automatically continue from where the job was ceased.")
        sys.exit(-1)

    # ---- prepare (MP:3320-3358): SAM/FASTA ingest replaces sam2bam / cat / sort / expand / strand split
    def _keep_regions(self, names, lens):
        """GFF masking (MP:817-859): keep regions as the reference's BED file, applied like `samtools view -L` on the combined records."""
        gff_ex, gff_in = self.opt.get("GFF_FILE_EXCLUDE", ""), self.opt.get("GFF_FILE_INCLUDE", "")
        regions = None
        if gff_ex and os.path.exists(gff_ex):
            self._say("Removing reads that are overlapped with features in the GFF file.")
            regions = gffmask.keep_regions_exclude(gff_ex, {n: int(l) for n, l in zip(names, lens)}, 55)
            if not regions:
                self._say("!!! No regions need to analyze after excluding regions in the GFF file, stop analyze!")
                sys.exit(-1)
        elif gff_in and os.path.exists(gff_in):
            self._say("GFF_FILE_INCLUDE specified, removing reads that are not overlap with features in the GFF file.")
            regions = gffmask.keep_regions_include(gff_in, 55)
            if not regions:
                self._say("!!! No regions in the GFF_FILE_INCLUDE file or all regions are shorter than 55, stop analyze!")
                sys.exit(-1)
        if regions is not None:
            regions = gffmask.regions_by_tid(regions, names)
        return regions

    def run_prepare(self):
        """One rank: ingest everything.  Several ranks with their own RCCL communicator: every rank tokenizes its byte range of every SAM file, the
        records travel to the rank that owns their contig (mirp_ingest_sams_shard) and every rank keeps `prepared_<rank>.npz`.  Ranks without a
        communicator (or compressed inputs): rank 0 ingests, the others wait on a token FILE, not inside a collective (a cfg5-sized ingest outlasts
        a collective's watchdog), with a liveness check on rank 0."""
        paths = self.opt["ALIGNMENT_FILE"]
        if getattr(self, "rccl", False) and not any(str(p).endswith(".gz") for p in paths):
            return self._prepare_sharded()
        token = self._all_gather("%d.%d" % (os.getpid(), int(time.time() * 1e3)))[0]      # rank 0's token names this run
        done, failed, beat = self._p("prepare.done." + token), self._p("prepare.failed." + token), self._p("prepare.alive." + token)
        if self.rank != 0:
            # the temporary folder must be shared by all ranks (one node, or a shared file system).  Rank 0 touches the heartbeat file while it
            # works; a rank 0 that died (OOM kill, SIGKILL) stops doing so and the waiting ranks leave instead of waiting forever.
            t_start = time.time()
            while not (os.path.exists(done) or os.path.exists(failed)):
                time.sleep(0.2)
                try:
                    age = time.time() - os.path.getmtime(beat)
                except OSError:
                    age = time.time() - t_start
                if age > float(os.environ.get("MIRP_PREPARE_TIMEOUT", "600")):
                    sys.stderr.write("Error: rank 0 gave no sign of life for %.0f s during the prepare stage (is %s shared by all ranks?); stopping.\n" % (age, self.tmp))
                    sys.exit(-1)
            if os.path.exists(failed):
                sys.exit(-1)
            return
        import threading
        stop = threading.Event()

        def heartbeat():
            while not stop.is_set():
                open(beat, "w").close()
                stop.wait(5.0)
        hb = threading.Thread(target=heartbeat, daemon=True)
        hb.start()
        try:
            self._prepare_rank0()
        except SystemExit:
            open(failed, "w").close()
            raise
        except BaseException:
            open(failed, "w").close()
            raise
        finally:
            stop.set()
            hb.join()
        for old in os.listdir(self.tmp):
            if old.startswith("prepare.done.") or old.startswith("prepare.failed.") or old.startswith("prepare.alive."):
                os.remove(os.path.join(self.tmp, old))
        open(done, "w").close()

    def _prepare_sharded(self):
        self._say("Starting preparing data for the 'candidate' stage.")
        paths = self.opt["ALIGNMENT_FILE"]
        names, lens = ingest.read_sam_header(paths[0])
        regions = self._keep_regions(names, lens)
        owner = np.zeros(len(names), dtype=np.int32)
        for r, part in enumerate(dist.partition_contigs(lens, self.world)):
            owner[part] = r
        ok, res = True, None
        try:
            res = self.ctx.ingest_sams_shard(paths, owner, regions=regions)
        except ValueError as e:
            sys.stderr.write(str(e) + "\n")
            ok = False
        self._agree_ok(ok, "prepare")
        names, lens, samples, alns, segs, self.ingest_seconds = res
        prepared = self._p("prepared_%d.npz" % self.rank)
        np.savez(prepared, contig_names=np.array(names, dtype=object), contig_lens=lens, sample_names=np.array(samples, dtype=object), alns=alns,
                 segs=segs, allow_pickle=True)
        files = self._all_gather(prepared)
        if self.rank == 0:
            d = {"last_stage": "prepare", "finished_stages": {"prepare": {"preparedname": files}}, "files": {"prepare": files}, "world": self.world}
            _save_recover(self.recovername, d)
        self._say("Done (prepare stage)\n")
        self._barrier()

    def _prepare_rank0(self):
        _msg("Starting preparing data for the 'candidate' stage.")
        paths = self.opt["ALIGNMENT_FILE"]
        names, lens = ingest.read_sam_header(paths[0])
        regions = self._keep_regions(names, lens)
        try:
            if any(str(p).endswith(".gz") for p in paths):        # compressed inputs: host parser (same rules), host filter and sort
                names, lens, samples, alns, segs = ingest.read_sams(paths, regions=regions, with_segments=True)
            elif early.has_ingest(paths):      # the CLI started the tokenizer while the device was being opened: the device half (filter, sort) follows now
                names, lens, samples, alns, segs, self.ingest_seconds = self.ctx.ingest_tokenized(paths, regions=regions)
            else:       # tokenizer on the host threads; keep-region filter and the stable (tid, pos) sort on the GPU (mirp_ingest_sams_gpu)
                names, lens, samples, alns, segs, self.ingest_seconds = self.ctx.ingest_sams(paths, regions=regions)
        except ValueError as e:
            sys.stderr.write(str(e) + "\n")
            sys.exit(-1)
        if getattr(self, "lean", False):          # handed to the candidate stage in memory (_load_inputs)
            self._prepared = {"contig_names": names, "contig_lens": lens, "sample_names": samples, "alns": alns, "segs": segs}
            _msg("Done (prepare stage)\n")
            return
        prepared = self._p("prepared.npz")
        np.savez(prepared, contig_names=np.array(names, dtype=object), contig_lens=lens, sample_names=np.array(samples, dtype=object), alns=alns,
                 segs=segs, allow_pickle=True)
        d = {"last_stage": "prepare", "finished_stages": {"prepare": {"preparedname": prepared}}, "files": {"prepare": [prepared]}, "world": self.world}
        _save_recover(self.recovername, d)
        _msg("Done (prepare stage)\n")

    def _load_inputs(self):
        if self.data is not None:
            return
        if self._prepared is not None:          # lean run: the prepare stage's result, still in memory
            z, sharded = self._prepared, False
            self._prepared = None
        else:
            d = load_recover_file(self.recovername)
            if d.get("world", 1) != self.world:      # piece files and contig shards are laid out per rank: the stages of one run share one world size
                if self.rank == 0:
                    sys.stderr.write("Error: the stage files in %s were written by a run with %d rank(s); this run has %d. Run the stages with the same "
                                     "number of ranks, or start again from 'prepare'.\n" % (self.tmp, d.get("world", 1), self.world))
                sys.exit(-1)
            prep = d["finished_stages"]["prepare"]["preparedname"]
            sharded = isinstance(prep, (list, tuple))          # one file per rank, each with the records of the rank's own contigs
            z = np.load(prep[self.rank] if sharded else prep, allow_pickle=True)
            z = {k: z[k] for k in z.files}
        names = [str(x) for x in z["contig_names"]]
        lens = z["contig_lens"]
        mine = np.ones(len(names), dtype=bool)
        if self.world > 1:          # contig sharding: whole contigs per rank, balanced by length
            mine[:] = False
            mine[dist.partition_contigs(lens, self.world)[self.rank]] = True
        # genome: only this rank's contigs are read and uploaded; the others stay in the contig table with length 0 (tids are genome-wide)
        fasta = self.opt["FASTA_FILE"]
        if str(fasta).endswith(".gz"):
            fa = dict(ingest.read_fasta(fasta))
            have = set(fa)
        else:
            got = capi.read_fasta(fasta, want=[n for n, m_ in zip(names, mine) if m_] if self.world > 1 else None)
            have = set(n for n, _ in got)
            fa = {n: a for n, a in got if a is not None}
        missing = [n for n in names if n not in have]
        if missing:          # every rank reads the same files and leaves together
            if self.rank == 0:
                sys.stderr.write("Error: sequence %s of the SAM header is not in the FASTA file\n" % missing[0])
            sys.exit(-1)
        alns = z["alns"]
        segs = z["segs"] if "segs" in z else alns[:0]
        if self.world > 1 and not sharded:
            alns = alns[mine[alns["tid"]]]
            segs = segs[mine[segs["tid"]]]
        empty = np.zeros(0, dtype=np.uint8)
        self.data = {"names": names, "lens": lens, "samples": [str(x) for x in z["sample_names"]], "alns": alns,
                     "contigs": [(n, fa[n] if m_ else empty) for n, m_ in zip(names, mine)]}
        self.ctx.load_genome(self.data["contigs"])
        self.ctx.load_alignments(self.data["alns"])
        if len(segs):
            self.ctx.load_coverage_segments(segs)

    def _ensure_candidate(self):
        if self.state in ("candidate", "fold"):
            return
        self._load_inputs()
        order = np.argsort(np.array(self.data["names"], dtype=object), kind="stable").astype(np.int32)  # sorted(dict_contigs), MP:1309
        self.counts = self.ctx.candidate(self.opt["READS_DEPTH_CUTOFF"], self.opt["MAX_GAP"], self.opt["PRECURSOR_LEN"], order)
        if self.world > 1:
            # the strand vote double-counts the first position of every contig's first run except the first run of the whole depth file
            # (MP:905-906, 926-929): a shard whose first covered contig is not the genome's first covered contig must do so for its first run too
            depth = self.ctx.get_depth()
            first = int(depth[0]["tid"]) if len(depth) else 1 << 30
            firsts = self._all_gather(first)
            if first != (1 << 30) and first != min(firsts):
                self.ctx.set_contig_shard(True)
                self.counts = self.ctx.candidate(self.opt["READS_DEPTH_CUTOFF"], self.opt["MAX_GAP"], self.opt["PRECURSOR_LEN"], order)
        self.state = "candidate"
        self._imported, self._moves = [], None          # window-level re-balancing happens once per candidate state, in front of the fold

    def _balance(self):
        """Several ranks: even out the window lists before the fold (balance.py).  Collective; every rank computes the same plan from the counts.
        With -d the run keeps whole contigs per rank (the reasons file and the failed read-mapping layouts are written from the owner's state)."""
        if self.world == 1 or self._moves is not None:
            return
        self._moves = []
        if self.opt.get("OUTPUT_DETAILS_FOR_DEBUG"):
            return
        counts = self._all_gather(int(self.counts[2]))
        if self.ctx.dist_world() == self.world:          # the context's communicator: RCCL, or the local transport of ranks that share a GPU
            xchg = self.ctx.exchange_bytes
        else:                                              # no library communicator (MIRP_DIST_BACKEND=gloo): host objects
            def xchg(blocks):
                return [b[self.rank] for b in self._all_gather(blocks)]
        ok, res = True, None
        try:
            res = balance.exchange(xchg, self.rank, self.world, self.ctx.get_windows, self.data["alns"], counts)
        except Exception as e:          # any failure of one rank (library error, a payload that does not unpack) must reach the agreement below:
            sys.stderr.write("%s: %s\n" % (type(e).__name__, e))      # a rank that died here would leave the others inside the next collective
            ok = False
        self._agree_ok(ok, "window re-balancing")
        keep, self._imported, self._moves = res
        if keep != int(self.counts[2]):
            self.ctx.limit_windows(keep)
        if self._moves and self.rank == 0:
            _msg("Re-balancing the fold: %s windows per rank, %d transfer(s) of windows between ranks." % (counts, len(self._moves)))

    # ---- candidate (MP:3361-3438)
    def run_candidate(self, defer=False):
        """defer (the `pipeline` verb on one rank): the Python-side artefacts of the stage -- the pickled dict_loci, ExRegionA.gff3, the window
        dump and the checkpoint record -- are written by a host thread behind the fold kernels (the device call releases the interpreter
        lock); run_fold joins it once the fold is done.  Ranks > 1 exchange objects in that code, which must stay on the main thread."""
        if self._prepared is None and not previous_stage_saved(self.recovername, "prepare"):
            self._fail_stage()
        self._say("Starting identifying candidate regions")
        self.state = None
        self._ensure_candidate()
        if self.lean:          # no artefact of this stage outlives the run: the windows stay on the device for the fold
            sys.stdout.write("%d candidate loci generated, %d regions to fold.\n" % (int(self.counts[1]), int(self.counts[2])))
            self._say("Done (candidate stage)\n")
            return
        names, prefix, r = self.data["names"], self.opt["NAME_PREFIX"], self.rank
        depthname = self._p("bam.depth.cut%d" % self.opt["READS_DEPTH_CUTOFF"])
        if self.world == 1:
            self.ctx.write_depth_text(depthname, names)          # formatted by the library's worker threads
        else:       # every rank writes the lines of its contigs; rank 0 stitches the parts back together in @SQ order
            depth = self.ctx.get_depth()
            part = depthname + ".part%d" % r
            spans = {}
            with open(part, "w") as f:
                tids = np.unique(depth["tid"]) if len(depth) else []
                lo = np.searchsorted(depth["tid"], tids, side="left") if len(depth) else []          # depth lines are in (tid, pos) order
                hi = np.searchsorted(depth["tid"], tids, side="right") if len(depth) else []
                for t, i0, i1 in zip(tids, lo, hi):
                    a = f.tell()
                    f.write(records.depth_text(depth[i0:i1], names))
                    spans[int(t)] = (a, f.tell())
            all_spans = self._all_gather(spans)
            if r == 0:
                with open(depthname, "w") as fo:
                    for t in range(len(names)):
                        for rr, sp in enumerate(all_spans):
                            if t in sp:
                                with open(depthname + ".part%d" % rr) as fi:
                                    fi.seek(sp[t][0])
                                    fo.write(fi.read(sp[t][1] - sp[t][0]))
        loci, psorted = self.ctx.get_loci()
        w = self.ctx.get_windows()
        fastaname = self._p(prefix + ".rnalfold.in_%d.fa" % r)       # one piece per rank, like the reference's pieces per process
        self.ctx.write_window_fasta(fastaname, names)                # headers (MP:1124-1140) + sequences, formatted by the library's worker threads
        self._cand_fasta = fastaname
        counts = (int(self.counts[1]), int(self.counts[2]))
        if defer and self.world == 1:
            self._cand_error = None

            def job():
                try:
                    self._candidate_artifacts(loci, psorted, w, depthname, fastaname, counts)
                except BaseException as e:          # surfaces in _join_candidate, on the main thread
                    self._cand_error = e
            self._cand_thread = threading.Thread(target=job)
            self._cand_thread.start()          # "Done (candidate stage)" is said by _join_candidate, once the artefacts and the checkpoint record exist
        else:
            self._candidate_artifacts(loci, psorted, w, depthname, fastaname, counts)
            self._say("Done (candidate stage)\n")
        self._barrier()

    def _join_candidate(self):
        t = getattr(self, "_cand_thread", None)
        if t is not None:
            t.join()
            self._cand_thread = None
            if self._cand_error is not None:
                raise self._cand_error
            self._say("Done (candidate stage)\n")

    def _candidate_artifacts(self, loci, psorted, w, depthname, fastaname, counts):
        names, prefix, r = self.data["names"], self.opt["NAME_PREFIX"], self.rank
        lociname = self._p(prefix + "_loci_dump.dump")
        dict_loci = {}
        for d_ in self._all_gather(records.loci_to_dict(loci, psorted, names, self.opt["PRECURSOR_LEN"])):
            dict_loci.update(d_)
        if r == 0:
            with open(lociname, "wb") as f:
                pickle.dump(dict_loci, f, protocol=2)
        # <prefix>_ExRegionA.gff3 (MP:1357-1369): every extended region with the peaks of its locus, numbered per contig
        exname = self._p(prefix + "_ExRegionA.gff3")
        if r == 0:
            with open(exname, "w") as f:
                f.write(records.exregion_gff_text(dict_loci))
        dumpname = self._p(prefix + ".alndump_%d.npz" % r)
        np.savez(dumpname, windows=w["windows"], wpeaks=w["wpeaks"], matures=w["matures"])
        parts = self._all_gather((fastaname, dumpname, counts[0], counts[1]))
        nloci, nfasta = sum(x[2] for x in parts), sum(x[3] for x in parts)
        if r == 0:
            d = load_recover_file(self.recovername)
            d["last_stage"] = "candidate"
            d["finished_stages"]["candidate"] = {"depthfilename": depthname, "loci_dump_name": lociname, "fasta": [x[0] for x in parts],
                                                 "infodump": [x[1] for x in parts], "num_loci": nloci, "num_fasta": nfasta}
            d["files"]["candidate"] = [x[0] for x in parts] + [x[1] for x in parts]
            _save_recover(self.recovername, d)
            sys.stdout.write("%d candidate loci generated, %d regions to fold.\n" % (nloci, nfasta))

    def _fold_device(self):
        """Fold every window on the device; returns the per-window status array.  A window can produce more structure lines than the
        default capacity of 96 (tandem repeats do: one line per start position is possible); RNALfold has no such limit (MP:3053), so
        mirp_fold folds just those windows again at the capacity no window can exceed, into side buffers the later stages read."""
        self._balance()
        self.ctx.fold(self.opt["PRECURSOR_LEN"])
        status = self.ctx.fold_status()
        for p in self._imported:          # windows received from over-loaded ranks (balance.py): folded through the batch entry point
            try:
                balance.fold_imported(self.ctx, p, self.opt["PRECURSOR_LEN"])
            except Exception as e:
                sys.stderr.write("%s: %s\n" % (type(e).__name__, e))
                status = np.concatenate([status, np.array([-1], dtype=status.dtype)])
        return status

    # ---- fold (MP:3441-3495)
    def run_fold(self, write_text=True, defer=False):
        """defer (the `pipeline` verb): the RNALfold-format text is formatted and written behind the predict stage's device and report work;
        the fold stage is recorded in the checkpoint file once that file is complete (run_predict joins the writer before it records itself)."""
        if not (self.lean and self.state == "candidate") and getattr(self, "_cand_thread", None) is None and not previous_stage_saved(self.recovername, "candidate"):
            self._fail_stage()
        self._say("Starting folding candidate sequences.")
        self._ensure_candidate()
        if self.lean and self.state == "candidate":
            # lean run: the fold, the filter and the report files are ONE pipelined native call (mirp_fold_predict_report_stream, run by the predict
            # stage below): chunks of the window list are folded and filtered while a host thread writes the previous chunk's read-mapping files
            self.state = "fold"
            self._stream = True
            return
        status = self._fold_device()
        self._join_candidate()          # the candidate stage's host artefacts were being written behind the fold kernels (`pipeline` verb)
        self.state = "fold"
        prefix = self.opt["NAME_PREFIX"]
        foldname = self._p(prefix + "_rnalfoldoutput_%d" % self.rank)
        bad = np.nonzero(status != 0)[0]
        if len(bad):
            sys.stderr.write("Error occurred when folding sequences (window %d, status %d).\n" % (bad[0], status[bad[0]]))
        self._agree_ok(len(bad) == 0, "fold")
        if self.lean:          # the structure lines stay on the device for the filter; mirp_write_fold_text remains the export for every other way of running
            self._pending_fold = None
            self._say("Done (fold stage)\n")
            return
        if write_text:
            d = load_recover_file(self.recovername)
            self.ctx.write_fold_text(d["finished_stages"]["candidate"]["fasta"][self.rank], foldname, wait=not defer)
        else:
            open(foldname, "w").close()
        mine = [foldname]
        for p in self._imported:          # the helper's share of the fold artefact: the windows it folded for rank `src`
            part = foldname + ".from%d" % int(p["meta"][0])
            with open(part, "w") as f:
                f.write(balance.fold_text(p, self.data["names"]) if write_text else "")
            mine.append(part)
        foldnames = [x for part in self._all_gather(mine) for x in part]
        self._pending_fold = foldnames if defer else None
        if not defer:
            self._record_fold(foldnames)
        self._say("Done (fold stage)\n")
        self._barrier()

    def _record_fold(self, foldnames):
        if self.rank == 0:
            d = load_recover_file(self.recovername)
            d["last_stage"] = "fold"
            d["finished_stages"]["fold"] = {"foldnames": foldnames}
            d["files"]["fold"] = foldnames
            _save_recover(self.recovername, d)

    # ---- predict (MP:3498-3627): the loci list, gff3, fasta / ss / csv / html / stat / readmapping files
    def run_predict(self):
        """Every way out of the stage -- normal return, sys.exit of an agreed failure, an exception -- first joins the native writer thread of the fold
        text (mirp_write_fold_text_async) and the candidate stage's artefact thread: a detached writer racing the process exit would leave a
        truncated <prefix>_rnalfoldoutput file."""
        try:
            return self._run_predict()
        finally:
            t = getattr(self, "_cand_thread", None)
            if t is not None:
                t.join()
                self._cand_thread = None
            try:
                self.ctx.wait_text()
            except Exception:
                pass

    def _run_predict(self):
        pending = getattr(self, "_pending_fold", None)
        if self.lean and self.state == "fold":
            return self._run_predict_lean()
        if pending is None and not previous_stage_saved(self.recovername, "fold"):
            self._fail_stage()
        self._say("Starting predicting miRNAs.")
        if self.state != "fold":          # the `predict` verb in a process of its own: the fold is redone on the device (0.07 s), its status checked as run_fold does
            self._ensure_candidate()
            status = self._fold_device()
            badf = np.nonzero(status != 0)[0]
            if len(badf):
                sys.stderr.write("Error occurred when folding sequences (window %d, status %d).\n" % (badf[0], status[badf[0]]))
            self._agree_ok(len(badf) == 0, "fold")
            self.state = "fold"
        ns = len(self.data["samples"])
        out = self.ctx.predict(ns, self.opt["MIN_MATURE_LEN"], self.opt["MAX_MATURE_LEN"], self.opt["ALLOW_3NT_OVERHANG"], self.opt["ALLOW_NO_STAR_EXPRESSION"])
        bad = np.nonzero(out["status"] != 0)[0]
        if len(bad):      # a capacity of the filter kernel was exceeded (structure pieces / candidate matures of one window): never truncate silently
            sys.stderr.write("Error occurred when predicting miRNAs: window %d exceeds the capacity of the filter kernel (status %d).\n" % (bad[0], out["status"][bad[0]]))
        imp = []
        if len(bad) == 0:
            try:
                params = (ns, self.opt["MIN_MATURE_LEN"], self.opt["MAX_MATURE_LEN"], 1 if self.opt["ALLOW_3NT_OVERHANG"] else 0,
                          1 if self.opt["ALLOW_NO_STAR_EXPRESSION"] else 0, 55)
                imp = [balance.predict_imported(self.ctx, p, params) for p in self._imported]
            except Exception as e:
                sys.stderr.write("Error occurred when predicting miRNAs (imported windows): %s\n" % e)
                bad = [0]
        self._agree_ok(len(bad) == 0, "predict")
        def finish_fold():          # the fold stage's text file is being written behind this stage: complete it, then record the stage
            if getattr(self, "_pending_fold", None) is not None:
                self.ctx.wait_text()
                self._barrier()
                self._record_fold(self._pending_fold)
                self._pending_fold = None
        prefix, outdir = self.opt["NAME_PREFIX"], self.opt["OUTFOLDER"]
        if self.opt.get("OUTPUT_DETAILS_FOR_DEBUG"):          # -d: why the other regions are not miRNAs (MP:3532-3543)
            rec = self.ctx.predict_reasons(ns, self.opt["MIN_MATURE_LEN"], self.opt["MAX_MATURE_LEN"], self.opt["ALLOW_3NT_OVERHANG"],
                                           self.opt["ALLOW_NO_STAR_EXPRESSION"])
            w = self.ctx.get_windows()
            rname = os.path.join(outdir, prefix + "_reason_why_not_miRNA.txt")
            mine = rname if self.world == 1 else rname + ".part%d" % self.rank
            failed = write_reasons(mine, w, w["matures"], self.data["names"], rec, self.ctx.get_fold(), self.data["samples"],
                                   self.opt["MIN_MATURE_LEN"], self.opt["MAX_MATURE_LEN"], self.opt["ALLOW_3NT_OVERHANG"])
            if self.world > 1:
                self._barrier()
                if self.rank == 0:
                    with open(rname, "w") as fo:
                        for rr in range(self.world):
                            with open(rname + ".part%d" % rr) as fi:
                                fo.write(fi.read())
            fpay = locus_payloads(failed, dict(self.data["contigs"]), self.data["names"], self.data["alns"], self.data["samples"])
            gathered = self._all_gather((failed, fpay))
            if self.rank == 0 and os.path.getsize(rname):          # `if failed_reasons:` (MP:3534): the folder exists even when it stays empty
                failed = [m for part in gathered for m in part[0]]
                fpay = [x for part in gathered for x in part[1]]
                write_readmapping(failed, fpay, None, None, self.data["samples"], None, os.path.join(outdir, "failed_readmapping"))
        # the exchange step of the path (the reference's result queue, MP:2461-2499): the loci list of every rank on rank 0, in rank
        # order -- over the context's RCCL communicator straight from the device-resident result, or as host objects when the ranks have none.
        # The rank that owns a locus' contig also prepares what the report files need from the genome and the reads (locus_payloads).
        local = result_records(out, self.data["names"])
        for x in imp:
            local += result_records(x, self.data["names"])
        adjust_mature_star(local)
        if self._moves:
            # re-balanced run: a rank's list holds loci of contigs it does not own, so every rank sees the whole list and prepares the report
            # payloads of the loci on ITS contigs (it has their genome and reads); rank 0 puts them back in list order
            result = [m for part in self._all_gather(local) for m in part]
            mine_names = set(n for n, sq in self.data["contigs"] if len(sq))
            idx = [k for k, m in enumerate(result) if m[0] in mine_names]
            pay = locus_payloads([result[k] for k in idx], dict(self.data["contigs"]), self.data["names"], self.data["alns"], self.data["samples"])
            payloads = [None] * len(result)
            for idx_r, pay_r in self._all_gather((idx, pay)):
                for k, x in zip(idx_r, pay_r):
                    payloads[k] = x
            if self.rank != 0:
                result = []
        else:
            pay = locus_payloads(local, dict(self.data["contigs"]), self.data["names"], self.data["alns"], self.data["samples"])
        if self._moves:
            pass          # (result / payloads were assembled above)
        elif self.rccl:
            g = self.ctx.gather_loci(0)
            result = result_records(g, self.data["names"]) if self.rank == 0 else []
            adjust_mature_star(result)
            payloads = [x for part in self._all_gather(pay) for x in part]
        else:
            parts = self._all_gather((local, pay))
            result = [m for part in parts for m in part[0]]     # rank order, as pieces in the reference
            payloads = [x for part in parts for x in part[1]]
        if self.rank != 0:
            finish_fold()
            self._barrier()
            return []
        if not result:
            finish_fold()
            _msg("0 miRNA identified. No result files generated.")
            self._barrier()
            return result
        order = sorted(range(len(result)), key=lambda k: result[k][:10])      # resultlist.sort() of gen_gff_from_result (MP:2622) names the loci
        result = [result[k] for k in order]
        payloads = [payloads[k] for k in order]
        counts = np.stack([x["counts"] for x in payloads])
        rm_thread = write_readmapping(result, payloads, None, None, self.data["samples"], counts, os.path.join(outdir, "readmapping"), background=True)
        gffname = os.path.join(outdir, prefix + "_miRNA.gff3")
        maturename = os.path.join(outdir, prefix + "_miRNA.mature.fa")
        stemloopname = os.path.join(outdir, prefix + "_miRNA.precursor.fa")
        write_report_files(result, [x["pre"] for x in payloads], self.data["samples"], counts,
                           {"gff": gffname, "mature": maturename, "precursor": stemloopname, "ss": os.path.join(outdir, prefix + "_miRNA.precursor.ss"),
                            "csv": os.path.join(outdir, prefix + "_miRNA.detail.csv"), "html": os.path.join(outdir, prefix + "_miRNA.detail.html"),
                            "stat": os.path.join(outdir, "miRNA.stat.txt")})
        with open(self._p(prefix + "_miRNA.info.dump"), "wb") as f:
            pickle.dump(result, f)
        rm_thread.join()
        finish_fold()
        d = load_recover_file(self.recovername)
        d["last_stage"] = "predict"
        d["finished_stages"]["predict"] = {"gffname": gffname, "maturename": maturename, "stemloopname": stemloopname}
        d["files"]["predict"] = [gffname, maturename, stemloopname]
        _save_recover(self.recovername, d)
        _msg("The output files are in " + outdir)
        sys.stdout.write("%d miRNAs identified.\n" % len(result))
        _msg("Done (predict stage)\n")
        self._barrier()
        return result

    def _run_predict_lean(self):
        """One process, no -k, no -d: the filter's flat result goes straight into the native report writer (mirp_write_result_reports: swap, list order,
        read counts, readmapping/ and the seven report files); no Python object per locus, no result pickle, no checkpoint record."""
        ns = len(self.data["samples"])
        prefix, outdir = self.opt["NAME_PREFIX"], self.opt["OUTFOLDER"]
        mark = "\x00SEQ\x00"
        form = [p for taxon in ("Viridiplantae", "ALL") for p in _mirbase_form_text(mark, taxon).split(mark)]
        if getattr(self, "_stream", False):
            self._stream = False
            nwin = int(self.counts[2])
            chunks = int(os.environ.get("MIRP_STREAM_CHUNKS", "0")) or max(1, min(12, (nwin + 3000) // 6000))      # about 6,000 windows a chunk: 23 per CU
            params = (ns, self.opt["MIN_MATURE_LEN"], self.opt["MAX_MATURE_LEN"], 1 if self.opt["ALLOW_3NT_OVERHANG"] else 0,
                      1 if self.opt["ALLOW_NO_STAR_EXPRESSION"] else 0, 55)
            try:
                n, used, self.stream_device_s = self.ctx.fold_predict_report_stream(self.opt["PRECURSOR_LEN"], params, chunks, self.data["names"],
                                                                                   [sq for _, sq in self.data["contigs"]], self.data["alns"], self.data["samples"],
                                                                                   form, outdir, prefix)
            except capi.MirpError as e:
                sys.stderr.write(str(e) + "\n")
                sys.exit(-1)
            self._say("Done (fold stage)\n")
            self._say("Starting predicting miRNAs.")
            if n == 0:
                _msg("0 miRNA identified. No result files generated.")
                return []
            _msg("The output files are in " + outdir)
            sys.stdout.write("%d miRNAs identified.\n" % n)
            _msg("Done (predict stage)\n")
            return range(n)
        self._say("Starting predicting miRNAs.")
        out = self.ctx.predict_raw(ns, self.opt["MIN_MATURE_LEN"], self.opt["MAX_MATURE_LEN"], self.opt["ALLOW_3NT_OVERHANG"], self.opt["ALLOW_NO_STAR_EXPRESSION"])
        bad = np.nonzero(out["status"] != 0)[0]
        if len(bad):
            sys.stderr.write("Error occurred when predicting miRNAs: window %d exceeds the capacity of the filter kernel (status %d).\n" % (bad[0], out["status"][bad[0]]))
            sys.exit(-1)
        if len(out["result"]) == 0:
            _msg("0 miRNA identified. No result files generated.")
            return out["result"]
        res, _, _ = capi.write_result_reports(out["result"], out["text"], self.data["names"], [sq for _, sq in self.data["contigs"]], self.data["alns"],
                                              self.data["samples"], form, outdir, prefix)
        _msg("The output files are in " + outdir)
        sys.stdout.write("%d miRNAs identified.\n" % len(res))
        _msg("Done (predict stage)\n")
        return res

    def _mark(self, stage):
        c = getattr(self, "clock", None)
        if c is not None and c.path:
            tm = self.ctx.last_timings()
            sd = getattr(self, "stream_device_s", None)          # lean run: fold + filter ran inside the predict stage's one pipelined call
            dev = {"candidate": tm["coverage_ms"] + tm["candidate_rest_ms"], "fold": 0.0 if (sd or self.lean) else tm["fold_ms"],
                   "predict": (sd["fold_s"] + sd["predict_s"]) * 1e3 if sd else tm["predict_ms"]}
            c.mark(stage, device_ms=dev.get(stage, 0.0))

    def run_pipeline(self):
        self.run_prepare()
        self._mark("prepare")
        self.run_candidate(defer=not self.lean)
        self._mark("candidate")
        self.run_fold(defer=not self.lean)
        self._mark("fold")
        res = self.run_predict()
        self._mark("predict")
        return res

    def run_recover(self):
        """The `recover` verb (MP:3740-3774): continue after the last recorded stage; nothing recorded -> a message, no action.  Returns True
        when stages ran to the end (the caller then removes the temporary folder unless -k, as the reference does)."""
        d = load_recover_file(self.recovername)
        last = d["last_stage"] if d else None
        if last:
            self._say("The pipeline was stopped after stage '" + last + "'.")
        if last == "predict":
            if self.rank == 0:
                sys.stdout.write("*** The pipeline has been finished on the input. No action is performed.\n\n")
            return False
        if last not in STAGES:
            self._say("No recovery information found. Please run the pipeline in the 'pipeline' mode.\n")
            return False
        nxt = STAGES[STAGES.index(last) + 1:]
        if self.rank == 0:
            sys.stdout.write("*** Starting the pipeline from stage '%s'.\n" % nxt[0])
        for s in nxt:
            getattr(self, "run_" + s)()
        return True


def result_records(out, names):
    """`result` list of gen_miRNA_loci_nopredict (MP:2492-2496): [chr, fold_s, fold_e, mat_s, mat_e, star_s, star_e, ss, strand, has_star, exprinfo]."""
    res = []
    R = out["result"]
    col = [R[k].tolist() for k in ("tid", "fold_s", "fold_e", "mat_s", "mat_e", "star_s", "star_e", "strand", "has_star", "reserved", "total_depth_mature",
                                   "total_depth_star")]          # columns as Python ints once, not one numpy scalar per field and locus
    for tid, fs, fe, ms, me, s0, s1, strand, has_star, f, dm, dstar, ss in zip(*col, out["ss"]):
        info = {"total_depth_mature": dm, "total_depth_star": dstar}
        if f & 1:
            info["max_imperfect_star"] = 1 if f & 8 else 0   # presence and zero/non-zero are what the gff writer reads (MP:2631)
            info["imperfect_star_which"] = ((f >> 1) & 3) - 1
        res.append([names[tid], fs, fe, ms, me, s0, s1, ss, records.STRAND[strand], bool(has_star), info])
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


def write_report_files(resultlist, pre_list, samples, counts, paths):
    """gff3, mature / precursor FASTA, structure file, detail csv / html and miRNA.stat.txt of a SORTED result list in one native call
    (mirp_write_reports: four threads format the seven files).  The writers below state the same formats in Python; tests/test_host_cpu.py holds
    both against the reference's files."""
    names, tid_of, loci = [], {}, []
    for m in resultlist:
        t = tid_of.setdefault(m[0], len(names))
        if t == len(names):
            names.append(m[0])
        e = m[-1]
        over = 0
        if "max_imperfect_star" in e and e["max_imperfect_star"] != 0:
            over = 2 if e["imperfect_star_which"] == 2 else 1
        loci.append([t, m[1], m[2], m[3], m[4], m[5], m[6], 1 if m[8] == "-" else 0, 0 if e["total_depth_star"] == 0 else 1, over])
    mark = "\x00SEQ\x00"
    form = [p for taxon in ("Viridiplantae", "ALL") for p in _mirbase_form_text(mark, taxon).split(mark)]
    capi.write_reports(np.array(loci, dtype=np.int32).reshape(-1, 10), names, [m[7] for m in resultlist], pre_list, samples, counts, form, paths)


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


_RC = bytes.maketrans(b"ATGCU", b"UACGA")   # get_complement, MP:232-235


def _faidx(contigs, chrom, s, e_incl):
    """`samtools faidx chr:s-e` (1-based, inclusive) as the report writers use it: upper case, T -> U (MP:2570-2590, 2659-2661)."""
    return contigs[chrom][s - 1:e_incl].tobytes().decode().upper().replace("T", "U")


def _revcomp(seq):
    return seq.encode().translate(_RC)[::-1].decode()


def _seq(seqs, idx, m, s, e_incl):
    """Sequence chrom:s-e of locus idx for the report writers.  `seqs` is the genome ({name: uint8 array}) or, in a sharded run, the list of
    locus payloads (locus_payloads) that the ranks owning the contigs computed: every coordinate the writers ask for lies inside the precursor."""
    if isinstance(seqs, dict):
        return _faidx(seqs, m[0], s, e_incl)
    return seqs[idx]["pre"][s - m[1]:e_incl - m[1] + 1]


def locus_payloads(resultlist, contigs, names, alns, samples):
    """What the report files need from the genome and the reads of every locus of `resultlist`, computed where the locus' contig lives (a rank of
    a sharded run holds only its own contigs and records): the forward-strand precursor text, the read counts per sample (gen_mirna_info,
    MP:2644-2728) and the body of the locus' read-mapping file (gen_map_result, MP:2907-2959; native, mirp_report_readmapping).  Rank 0
    formats the files from these."""
    counts = mirna_read_counts(resultlist, names, alns, len(samples))
    bodies = readmapping_bodies(resultlist, contigs, names, alns, samples, counts)
    return [{"pre": _faidx(contigs, m[0], m[1], m[2] - 1), "counts": counts[idx], "map": bodies[idx]} for idx, m in enumerate(resultlist)]


def readmapping_bodies(resultlist, contigs, names, alns, samples, counts):
    tid_of = {n: t for t, n in enumerate(names)}
    loci = np.array([[tid_of[m[0]], m[1], m[2], m[3], m[4], m[5], m[6], 1 if m[8] == "-" else 0] for m in resultlist], dtype=np.int32).reshape(-1, 8)
    return capi.report_readmapping(loci, [m[7] for m in resultlist], alns, [contigs.get(n) for n in names], samples, counts[:, :, 0])


def write_fasta_ss(resultlist, contigs, maturename, stemloopname, ssname):
    """gen_mirna_fasta_ss_from_result (MP:2963-3019): mature / precursor FASTA and the structure file with its M/S annotation line."""
    with open(maturename, "w") as fm, open(stemloopname, "w") as fp, open(ssname, "w") as fs:
        for idx, m in enumerate(resultlist):
            mirname = "miRNA-precursor_%d" % idx
            matureid = ">%s:%d-%d %s %s" % (m[0], m[3], m[4] - 1, m[8], mirname)
            stemloopid = ">%s:%d-%d %s %s" % (m[0], m[1], m[2] - 1, m[8], mirname)
            matureseq = _seq(contigs, idx, m, m[3], m[4] - 1)
            stemloopseq = _seq(contigs, idx, m, m[1], m[2] - 1)
            ms, me, ss_, se = m[3] - m[1], m[4] - m[1], m[5] - m[1], m[6] - m[1]
            if ms < ss_:
                seq_dot = "." * ms + "M" * (me - ms) + "." * (ss_ - me) + "S" * (se - ss_) + "." * (len(stemloopseq) - se)
            else:
                seq_dot = "." * ss_ + "S" * (se - ss_) + "." * (ms - se) + "M" * (me - ms) + "." * (len(stemloopseq) - me)
            if m[8] == "-":
                matureseq, stemloopseq, seq_dot = _revcomp(matureseq), _revcomp(stemloopseq), seq_dot[::-1]
            fm.write(matureid + "\n" + matureseq + "\n")
            fp.write(stemloopid + "\n" + stemloopseq + "\n")
            fs.write(stemloopid + "\n" + stemloopseq + "\n" + m[7] + "\n" + seq_dot + "\n")


def mirna_read_counts(resultlist, names, alns, n_samples):
    """Per locus and sample: reads on the precursor / exactly the mature / exactly the star / antisense (gen_mirna_info, MP:2644-2728),
    from the position-sorted alignment records instead of one `samtools view` per locus.  -> int64 array [n_loci, n_samples, 4].
    All loci at once: the (locus, record) pairs of the loci's record ranges are flattened and binned."""
    n = len(resultlist)
    out = np.zeros((n, n_samples, 4), dtype=np.int64)
    if n == 0 or len(alns) == 0:
        return out
    tid_of = {nm: t for t, nm in enumerate(names)}
    L = np.array([[tid_of[m[0]], m[1], m[2], m[3], m[4], m[5], m[6], 1 if m[8] == "-" else 0] for m in resultlist], dtype=np.int64)
    key = alns["tid"].astype(np.int64) << 32 | alns["pos"].astype(np.int64)
    lo = np.searchsorted(key, (L[:, 0] << 32) | L[:, 1], side="left")
    hi = np.searchsorted(key, (L[:, 0] << 32) | L[:, 2], side="left")
    cnt = hi - lo
    tot = int(cnt.sum())
    if tot == 0:
        return out
    loc = np.repeat(np.arange(n), cnt)
    ridx = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt) + np.repeat(lo, cnt)
    a = alns[ridx]
    pos, ln, depth = a["pos"].astype(np.int64), a["len"].astype(np.int64), a["depth"].astype(np.int64)
    inside = pos + ln <= L[loc, 2]                                   # startpos >= locus_start and startpos + readlen <= locus_end
    sense = a["strand"].astype(np.int64) == L[loc, 7]
    smp = a["sample"].astype(np.int64)
    cat = [inside & sense, inside & sense & (pos == L[loc, 3]) & (ln == L[loc, 4] - L[loc, 3]), inside & sense & (pos == L[loc, 5]) & (ln == L[loc, 6] - L[loc, 5]),
           inside & ~sense]
    for c, msk in enumerate(cat):
        if msk.any():
            np.add.at(out, (loc[msk], smp[msk], c), depth[msk])
    return out


def write_readmapping(resultlist, contigs, names, alns, samples, counts, folder, background=False):
    """gen_map_result (MP:2907-2959): one <precursor id>.map.txt per locus with the reads of every sample laid out under the precursor.
    `contigs` = the genome dict (names / alns / counts are then used) or the list of locus payloads of a sharded run (bodies come ready).
    background: the files are written by a thread that is returned (join it) while the caller formats the other report files."""
    os.makedirs(folder, exist_ok=True)
    if isinstance(contigs, dict):
        bodies = readmapping_bodies(resultlist, contigs, names, alns, samples, counts)
    else:
        bodies = [x["map"] for x in contigs]
    jobs = []
    for idx, m in enumerate(resultlist):
        mirname = "miRNA-precursor_%d" % idx
        jobs.append((os.path.join(folder, mirname + ".map.txt"), ">%s %s:%d-%d %s\n" % (mirname, m[0], m[1], m[2], m[8]) + bodies[idx]))

    def put_all():          # thousands of small files: creating them is the cost; done natively, outside the interpreter lock (mirp_write_files)
        capi.write_files([j[0] for j in jobs], [j[1] for j in jobs])
    if not background:
        put_all()
        return None
    import threading

    class Writer(threading.Thread):          # join() re-raises what the writer hit: a report folder with files missing is not a finished stage
        error = None

        def run(self):
            try:
                put_all()
            except BaseException as e:
                self.error = e

        def join(self, timeout=None):
            threading.Thread.join(self, timeout)
            if self.error is not None:
                raise self.error
    th = Writer()
    th.start()
    return th


def locus_texts(resultlist, contigs):
    """Per locus: (precursor, mature, star) as the detail tables print them (reverse complement on the minus strand), plus the length and first base of
    the forward-strand mature (gen_miRNA_stat, MP:2731-2741).  The csv and the html table print the same texts: run_predict builds them once and
    hands them to both writers (`texts`)."""
    out = []
    for idx, m in enumerate(resultlist):
        pre = _seq(contigs, idx, m, m[1], m[2] - 1)
        mat = _seq(contigs, idx, m, m[3], m[4] - 1)
        star = _seq(contigs, idx, m, m[5], m[6] - 1)
        flen, ffirst = len(mat), mat[0]
        if m[8] == "-":
            pre, mat, star = _revcomp(pre), _revcomp(mat), _revcomp(star)
        out.append((pre, mat, star, flen, ffirst))
    return out


def write_csv_and_stat(resultlist, contigs, samples, counts, csvname, statname, texts=None):
    """gen_csv_table (MP:2744-2779) and the miRNA.stat.txt block of the predict stage (MP:3585-3593)."""
    dict_len, dict_first = {}, {}
    if texts is None:
        texts = locus_texts(resultlist, contigs)
    cstr = [[[str(v) for v in per] for per in row] for row in np.asarray(counts).astype(np.int64).tolist()] if len(resultlist) else []
    with open(csvname, "w") as f:
        head = "miRNAID, Seqid(chromosome), start position, end position, strand, precursor sequence, secondary structure, mature sequence, star sequence, "
        f.write(head + "".join(s + "," + s + "," + s + "," + s + "," for s in samples) + "\n")
        f.write(head + "reads mapped to precursor, reads mapped to mature, reads mapped to star, reads mapped to antisense region," * len(samples) + "\n")
        lines = []
        for idx, m in enumerate(resultlist):
            pre, mat, star, flen, ffirst = texts[idx]
            dict_len[flen] = dict_len.get(flen, 0) + 1                     # gen_miRNA_stat (MP:2731-2741) uses the forward-strand text
            dict_first[ffirst] = dict_first.get(ffirst, 0) + 1
            row = ["miRNA-precursor_%d" % idx, m[0], str(m[1]), str(m[2]), m[8], pre, m[7], mat, star]
            for s in range(len(samples)):
                row += cstr[idx][s]
            lines.append(", ".join(row))
        if lines:
            f.write("\n".join(lines) + "\n")
    with open(statname, "w") as f:
        f.write("Total number of predicted miRNAs: %d\n" % len(resultlist))
        f.write("Distribution of the length of the mature miRNAs:\n")
        for k in sorted(dict_len):
            f.write("%s: %d\n" % (k, dict_len[k]))
        f.write("Distribution of the nucleotide of the first base of the mature miRNAs:\n")
        for k in sorted(dict_first):
            f.write("%s: %d\n" % (k, dict_first[k]))


_MIRBASE_PARTS = {}


def _mirbase_form(seq, taxon):
    """The miRBase BLAST search form the reference attaches to every mature sequence (gen_search_miRBase_str, MP:2773-2791); everything but the
    sequence is the same for a taxon and is put together once."""
    ps = _MIRBASE_PARTS.get(taxon)
    if ps is None:
        mark = "\x00SEQ\x00"
        whole = _mirbase_form_text(mark, taxon)
        ps = _MIRBASE_PARTS[taxon] = whole.split(mark)
    return seq.join(ps)


def _mirbase_form_text(seq, taxon):
    hidden = [("sequence", " " + seq), ("seqfile", ""), ("type", "mature"), ("search_method", "blastn"), ("evalue", "10"), ("maxalign", "100"),
              ("taxon", " " + taxon)]
    gap = " " * 5
    parts = ['<form name="myform" action="http://www.mirbase.org/cgi-bin/blast.pl" method="POST">', '<div align="center">']
    parts += ['<input type="hidden" size="25" name="%s" value="%s">' % kv for kv in hidden]
    parts += ['<input type="submit" value="Search mature on miRBase ( %s )" onclick="this.form.target=\'_blank\';return true;">' % taxon, "</div>", "</form>"]
    return gap.join(parts)


def write_html(resultlist, contigs, samples, counts, htmlname, texts=None):
    """gen_html_table_file (MP:2793-2904): the two distribution tables of miRNA.stat.txt and the detail table of the csv as markup, with a miRBase
    search form per mature sequence and a link to the locus' read-mapping file."""
    dict_len, dict_first = {}, {}
    rows = []
    if texts is None:
        texts = locus_texts(resultlist, contigs)
    cstr = [[[str(v) for v in per] for per in row] for row in np.asarray(counts).astype(np.int64).tolist()] if len(resultlist) else []
    for idx, m in enumerate(resultlist):
        pre, mat, star, flen, ffirst = texts[idx]
        dict_len[flen] = dict_len.get(flen, 0) + 1                         # forward-strand text, as in the stat file
        dict_first[ffirst] = dict_first.get(ffirst, 0) + 1
        name = "miRNA-precursor_%d" % idx
        cells = [name, m[0], str(m[1]), str(m[2]), m[8]]
        tail = [mat + _mirbase_form(mat, "Viridiplantae") + _mirbase_form(mat, "ALL"), star]
        for s in range(len(samples)):
            tail += cstr[idx][s]
        td = "\t\t\t<td nowrap>%s</td>\n"
        rows.append("\t\t<tr>\n" + "".join(td % c for c in cells) + "\t\t\t<td nowrap> <code>" + pre + "<BR>" + m[7] + " </code></td>" +
                    "".join(td % c for c in tail) + '\t\t\t<td><a href="readmapping/' + name + '.map.txt" target="_blank">Click to see detailed mapping.</a></td>' +
                    "\t\t</tr>\n")

    def dist_table(title, head, dist):
        out = "<h3>" + title + "</h3>\n<table border=\"1\">\n\t<thead>\n\t\t<tr>\n\t\t<th>" + head + "</th>\n\t\t<th>Count</th>\n\t\t</tr>\n\t</thead>\n\t<tbody>\n"
        for k in sorted(dist):
            out += "\t\t<tr>\n\t\t\t<td>%s </td>\n\t\t\t<td>%d </td>\n\t\t</tr>\n" % (k, dist[k])
        return out + "\t</tbody>\n</table>\n</div>"

    colors = ["#A9E2F3", "#ACFA58", "#F5A9BC"]
    head1 = ["miRNA precursor ID", "Chromosome", "start position", "end position", "strand", "precursor sequence and secondary structure",
             "mature sequence", "star sequence"]
    per_sample = ["precursor", "mature", "star", "antisense region"]
    with open(htmlname, "w") as f:
        f.write("<h1 > microRNAs predicted by miR-PREFeR </h1>\n<div>\n<h2 > Total number of prediction:%d  </h2>\n" % len(resultlist))
        f.write(dist_table("Distribution of the lengths of the mature sequences", "Length", dict_len) + "\n")
        f.write(dist_table("Distribution of the nucleotide of the first base of the mature sequences", "Nucleotide", dict_first))
        f.write("<div><h3>Detailed infomation </h3>\n<table border=\"1\">\n<colgroup>\n\t<col span=8 style=\"background-color:#CECEF6\">\n")
        f.write("".join('\t<col span="4" style="background-color:%s">' % colors[i % len(colors)] for i in range(len(samples))) + "</colgroup>\n")
        f.write("\t<thead>\n\t\t<tr>\n" + "".join('\t\t\t<th rowspan="2">%s</th>\n' % c for c in head1))
        f.write("".join('\t\t\t<th colspan="4">%s</th>\n' % s for s in samples) + '\t\t\t<th colspan="2">read mappings</th>\n\t\t</tr>\n\t\t<tr>\n')
        f.write("".join("\t\t\t<th> reads mapped to %s </th>\n" % w for _ in samples for w in per_sample) + "\t\t</tr>\n\t</thead>\n\t<tbody>\n")
        f.write("".join(rows) + "\t</tbody>\n</table>\n</div>\n")


# ---- -d artefact: <prefix>_reason_why_not_miRNA.txt (convert_failure_reasons_list MP:2505-2529, write_dict_reasons MP:2532-2567)
MS_FAIL = {1: "FAIL_STRUCTURE_MATCHED_BASES", 2: "FAIL_STRUCTURE_MATURE_NOT_IN_FOLD_REGION", 3: "FAIL_STRUCTURE_MATURE_NOT_IN_ONE_ARM",
           4: "FAIL_STRUCTURE_MATURE_MATCH_SMALL_THAN_14", 5: "FAIL_STRUCTURE_MATURE_STAR_OVERLAP", 6: "FAIL_STRUCTURE_STAR_OUT_OF_FOLD_REGION",
           7: "FAIL_STRUCTURE_STAR_NOT_IN_ONE_ARM", 8: "FAIL_STRUCTURE_TOO_MANY_BULGE_OR_LOOP", 9: "FAIL_STRUCTURE_MAX_BULGE_LARGE_THAN_2",
           10: "FAIL_STRUCTURE_TOTAL_LOOP_SIZE_LARGER_THAN_5", 11: "FAIL_STRUCTURE_NUM_BULGE_MORE_THAN_2"}


def _expression_info_lines(r, samples, allow_3nt):
    """The non-dict items of check_expression_new's result in insertion order (MP:2113-2163), `key\tstr(value)`."""
    ns = len(samples)
    this, anti, mature, iso, star = (int(r[k]) for k in (12, 13, 14, 15, 16))
    imp = [int(r[17]), int(r[18]), int(r[19])]
    out = [("samplenames", str(list(samples))), ("total_depth_just_this_strand", this), ("total_depth_anti", anti), ("total_depth_mature", mature),
           ("total_depth_isoform", iso), ("total_depth_star", None), ("total_depth_imperfect_star", str(imp)),
           ("mature_depth_each_sample", str([int(x) for x in r[21:21 + ns]])), ("mature_star_distance", int(r[20]))]
    max_imp = 0
    has_key = star == 0 and allow_3nt
    if has_key:
        max_imp = max(imp)
        if max_imp == 0:
            out += [("imperfect_star_start", 0), ("max_imperfect_star", 0)]
        else:
            which = imp.index(max_imp)
            s0, s1 = int(r[10]), int(r[11])
            st, en = [(s0, s1 + 1), (s0 + 1, s1), (s0 + 1, s1 + 1)][which]
            out += [("imperfect_star_start", st), ("imperfect_star_end", en), ("max_imperfect_star", max_imp), ("imperfect_star_which", which)]
    top = max(star, max_imp)
    out += [("mature_star_ratio_total", float(mature + top) / this), ("mature_star_ratio_total_both_strand", float(mature + top) / (this + anti)),
            ("mature_iso_star_ratio_total", float(iso + top) / this)]
    final_star = top if has_key else star
    return ["%s\t%s" % (k, final_star if k == "total_depth_star" else v) for k, v in out]


def write_reasons(path, windows, matures, names, records_arr, fold_raw, samples, min_mature_len, max_mature_len, allow_3nt):
    """Text of write_dict_reasons for the regions that produced no miRNA.  Block order: contig by first appearance, '+' before '-', regions
    by first appearance (the order of Python dicts under the reference's py3 shim; the py2 original iterates in hash order).
    -> the `ss_info` entries of the (mature, structure) pairs that reached the expression test and failed it, in file order: the
    list the reference hands to gen_mirna_info / gen_map_result for the `failed_readmapping` folder (MP:2561-2567)."""
    win = windows["windows"]
    mat = matures
    per_window, pairs = {}, {}
    for r in records_arr:
        if r[1] < 0:
            per_window[int(r[0])] = r
        else:
            pairs.setdefault(int(r[0]), []).append(r)
    emitted = []          # (window index, which)
    i, nw = 0, len(win)
    while i < nw:
        if win[i]["tag"] == 0:
            if per_window[i][4] == 0:
                emitted.append((i, "0"))
            i += 1
        else:                                        # (L, R) pair of FASTA entries, paired by position (MP:2394-2403)
            if per_window[i][4] == 0:
                emitted.append((i, "L"))
                if i + 1 < nw and per_window[i + 1][4] == 0:
                    emitted.append((i + 1, "R"))
            i += 2
    blocks = {}           # chrom -> strand -> region -> text lines
    for w, which in emitted:
        W = win[w]
        chrom, strand = names[W["tid"]], records.STRAND[W["strand"]]
        region = (int(W["ws"]), int(W["we"]))
        peak = "%d-%d" % (W["loc_s"], W["loc_e"])
        head = ["%s:%d-%d\t%s\tpeak-region:%s\t%s" % (chrom, region[0], region[1], strand, peak, which), "PEAK_PASS_DEPTH:PASSED"]
        pw = per_window[w]
        tail, failed_here = [], []
        if pw[2] == 0:
            head.append("HAS_STEMLOOP_STRUCTURE:FAILED")
        else:
            head.append("HAS_STEMLOOP_STRUCTURE:PASSED")
            head.append("HAS_MATURE_SIZE_IN_RANGE:" + ("PASSED" if pw[3] else "FAILED"))
            if pw[3]:
                ms = mat[W["mature_off"]:W["mature_off"] + W["n_matures"]]
                order = sorted(range(len(ms)), key=lambda k: -int(ms[k]["depth"]))       # stable, depth descending (MP:2241)
                by_pair = {(int(r[1]), int(r[2])): r for r in pairs.get(w, [])}
                entries = {}                                                              # key1 -> lines (dict semantics: position of first insert, last value)
                nst = int(pw[2])
                for k in order:
                    m0, m1 = int(ms[k]["start"]), int(ms[k]["end"])
                    if m1 - m0 < min_mature_len or m1 - m0 > max_mature_len:
                        continue
                    for s in range(nst):
                        r = by_pair.get((k, s))
                        if r is None:
                            continue
                        wl, wss = capi.fold_window_lines(fold_raw, w)
                        ln = wl[r[3]]
                        if int(ln["energy"]) > 0:                                         # `if energy > lowest_energy: continue`, lowest stays 0 in a failing region
                            continue
                        ss = wss[r[3], r[4]:r[4] + r[5]].tobytes().decode()
                        lines, info = [], None
                        if r[6] != 0:
                            lines.append(MS_FAIL.get(int(r[6]), "FAIL_STRUCTURE_EXCEPTION") + "\tFAILED")
                        else:
                            f = int(r[7])
                            if f & 127:                                                   # MP:2281-2340: every expression failure keeps its ss_info
                                info = [chrom, int(r[8]), int(r[9]), m0, m1, int(r[10]), int(r[11]), ss, strand, not (f & (4 | 8))]
                            if f & 1: lines.append("FAIL_EXPRESS_PATTERN_MATURE_STAR_TOO_CLOSE\tFAILED")
                            if f & 2: lines.append("FAIL_EXPRESS_PATTERN_HAS_STAR_BUT_TOO_FEW_READS_MAPPED_TO_DUPLEX\tFAILED")
                            if f & 4: lines.append("FAIL_EXPRESS_PATTERN_NO_STAR_EXPRESSION_DISALLOW_NO_STAR\tFAILED")
                            if f & 8: lines.append("FAIL_EXPRESS_PATTERN_NO_STAR_EXPRESSION_TOO_MANY_START\tFAILED")
                            if f & 16: lines.append("FAIL_EXPRESS_PATTERN_NO_STAR_MATURE_STAR_RATIO_TOO_SMALL\tFAILED")
                            if f & 32: lines.append("FAIL_EXPRESS_PATTERN_NO_STAR_MATURE_DEPTH_TOO_SMALL\tFAILED")
                            if f & 64: lines.append("FAIL_EXPRESS_PATTERN_NO_STAR_MATURE_NOT_IN_ALL_SAMPLE\tFAILED")
                            if f & (4 | 8 | 16 | 32 | 64):
                                lines += _expression_info_lines(r, samples, allow_3nt)
                        entries[(m0, m1, strand, ss)] = (["MATURE region: %d-%d, SS: %s" % (m0, m1, ss)] + lines, info)
                for v, info in entries.values():
                    tail += v
                    if info is not None:
                        failed_here.append(info)
        body = head + ["which:" + which, "peak:" + peak] + tail
        blocks.setdefault(chrom, {"+": {}, "-": {}})[strand][region] = (body, failed_here)
    failed = []
    with open(path, "w") as f:
        for chrom in blocks:
            for strand in ("+", "-"):
                for region, (body, failed_here) in blocks[chrom][strand].items():
                    f.write("===========================================================\n")
                    f.write("\n".join(body) + "\n\n")
                    failed += failed_here
    return failed
