"""Command line of the MI355X host: same verbs, flags and config file as the reference
(/root/reference/miR_PREFeR.py:44-78 parse_option_optparse, verbs dispatched at :3728-3774).

    python -m mir_prefer_amd.cli [-L] [-k] [-d] [--device N] [--fold-model vienna-2.1.2|vienna-1.8.5]
                                 {check|prepare|candidate|fold|predict|pipeline|recover} configfile
"""
import logging
import optparse
import os
import shutil
import sys

ACTIONS = ["check", "prepare", "candidate", "fold", "predict", "pipeline", "recover"]


def parse_option_optparse(argv=None):
    usage = "Usage: %prog [options] action configfile\n\naction: one of " + ", ".join(ACTIONS)
    parser = optparse.OptionParser(usage)
    parser.add_option("-L", "--log", action="store_true", dest="log", help="Generate a log file.")
    parser.add_option("-k", "--keep-tmp", action="store_true", dest="keeptmp", help="After finish the whole pipeline, do not remove the temporary folder.")
    parser.add_option("-d", "--output-detail-for-debug", action="store_true", dest="debug", help="Output detailed information for debug.")
    parser.add_option("--device", type="int", default=0, help="GPU index (default 0).")
    parser.add_option("--fold-model", dest="fold_model", default="vienna-2.1.2", choices=["vienna-2.1.2", "vienna-1.8.5"],
                      help="Which RNALfold the fold stage reproduces: the reference runs whatever RNALfold is on PATH and bundles 2.1.2 "
                           "(Turner-2004, dangles 2; default) and 1.8.5 (Turner-1999, dangles 1).")
    options, args = parser.parse_args(argv)
    if len(args) != 2:
        parser.error("incorrect number of arguments. Run the script with -h option to see help.")
    if args[0] not in ACTIONS:
        parser.error("unknow command. Run the script with -h option to see help.")
    return {"action": args[0], "config": args[1], "log": bool(options.log), "keeptmp": bool(options.keeptmp), "debug": bool(options.debug),
            "device": options.device, "fold_model": options.fold_model}


class _Clock:
    """MIRP_CLI_TIMINGS=<file>: where the process spent its wall-clock, as JSON (bench.py's end-to-end leg reads it; times are time.time() stamps,
    so a parent can place its own spawn stamp in front).  Without the variable nothing is recorded."""

    def __init__(self):
        self.path = os.environ.get("MIRP_CLI_TIMINGS")
        self.marks = []

    def mark(self, name, **extra):
        if self.path:
            import time
            self.marks.append(dict(extra, name=name, t=time.time()))

    def dump(self):
        if self.path:
            import json
            with open(self.path, "w") as f:
                json.dump(self.marks, f)


def main(argv=None):
    clock = _Clock()
    clock.mark("main")
    o = parse_option_optparse(argv)
    from . import config
    opt = config.parse_configfile(o["config"])
    if o["action"] != "check" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # one process: open the device (0.2 - 0.3 s inside the kernel driver) and read the genome on threads of their own while this one imports numpy
        # and tokenizes the SAM files (early.py); capi.Context / capi.read_fasta adopt the results
        from . import early
        early.start_context(o["device"])
        if o["action"] == "pipeline" and opt["FASTA_FILE"]:
            early.start_fasta(opt["FASTA_FILE"])
        if o["action"] in ("pipeline", "prepare"):
            early.start_ingest(opt["ALIGNMENT_FILE"])          # the tokenizer (host half of mirp_ingest_sams_gpu); the device half follows in the prepare stage
    from . import capi, pipeline
    clock.mark("imports")
    opt["OUTPUT_DETAILS_FOR_DEBUG"] = o["debug"]
    if not opt["NAME_PREFIX"]:
        opt["NAME_PREFIX"] = "miR-PREFeR"
    os.makedirs(opt["OUTFOLDER"], exist_ok=True)
    if o["log"]:
        logging.basicConfig(filename=os.path.join(opt["OUTFOLDER"], opt["NAME_PREFIX"] + ".log"), filemode="w", level=logging.INFO,
                            format="%(asctime)20s  %(name)10s:  %(levelname)10s  %(message)s")
    if o["action"] == "check":
        # tool presence check of the reference (MP:3206-3269) becomes: is the HIP library built and a GPU usable?
        try:
            capi.load_library()
            ctx = capi.Context(o["device"])
            ctx.close()
            print("libmirprefer.so: OK (ABI %d); GPU %d: OK" % (capi.load_library().mirp_abi_version(), o["device"]))
        except capi.MirpError as e:
            sys.stderr.write(str(e) + "\n")
            sys.exit(-1)
        last = pipeline.detect_stage_last_finished(os.path.join(opt["TMPFOLDER"] or os.path.join(opt["OUTFOLDER"], opt["NAME_PREFIX"] + "_tmp"),
                                                                opt["NAME_PREFIX"] + "_recover"))
        print("Last finished stage: %s" % last)
        return 0
    # one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m mir_prefer_amd.cli ...`): contigs are sharded over the ranks.  Host
    # objects (file names, counts, report payloads) travel over a CPU-side `gloo` group; the data path's exchanges (record routing of the ingest,
    # gather of the loci list) run over RCCL on the library's own communicator, one rank per GPU (dist.init_context).  MIRP_DIST_BACKEND=gloo
    # keeps everything on the host channel (ranks that share one GPU, as in the tests).
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("MIRP_DIST_BACKEND", "rccl")
    if world > 1:
        from . import dist as mdist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        tdist = mdist.init_host_group()
        if backend not in ("gloo", "local"):
            o["device"] = int(os.environ.get("LOCAL_RANK", str(rank)))
    try:
        dctx = None
        if world > 1 and backend == "local":        # ranks that share one GPU: the library's exchanges go through files (mirp_dist_init_local)
            xdir = os.path.join(opt["TMPFOLDER"] or os.path.join(opt["OUTFOLDER"], opt["NAME_PREFIX"] + "_tmp"), "dist_xchg")
            if rank == 0:
                shutil.rmtree(xdir, ignore_errors=True)
                os.makedirs(xdir)
            tdist.barrier()
            dctx = capi.Context(o["device"])
            dctx.dist_init_local(xdir, rank, world)
        elif world > 1 and backend != "gloo":
            from . import dist
            dctx = dist.init_context(capi.Context(o["device"]), rank, world)
        p = pipeline.Pipeline(opt, o["device"], fold_model=o["fold_model"], rank=rank, world=world, ctx=dctx,
                              lean=o["action"] == "pipeline" and not o["keeptmp"])
        p.clock = clock
        clock.mark("context")
        def removetmp():                        # run_removetmp (MP:3630-3639): unless -k; DELETE_IF_SUCCESS is parsed but unused, as in the reference
            if not o["keeptmp"]:
                if world > 1:
                    import torch.distributed as tdist
                    tdist.barrier()
                if rank == 0:
                    pipeline._msg("Removing the temporary folder.")
                    shutil.rmtree(p.tmp, ignore_errors=True)
                    sys.stdout.write("Temporary folder removed.\n\n")
        if o["action"] == "pipeline":
            p.run_pipeline()
            clock.mark("stages")
            removetmp()
            clock.mark("removetmp")
        elif o["action"] == "recover":
            if p.run_recover():
                removetmp()
        else:
            getattr(p, "run_" + o["action"])()
    except capi.MirpError as e:
        sys.stderr.write(str(e) + "\n")   # reference behaviour for a failed tool: message on stderr, exit status -1
        sys.exit(-1)
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()
    clock.mark("end")
    clock.dump()
    return 0


if __name__ == "__main__":
    rc = main()
    # Leave without tearing the interpreter, numpy and the HIP runtime down piece by piece (0.04 s): every file is closed and every writer thread joined
    # by now (run_predict), and the kernel driver releases the device state of a process that ends either way (0.08 s of its own, not ours to shorten).
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(rc or 0)
