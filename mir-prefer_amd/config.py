"""`KEY=VALUE` configuration file: same keys, defaults and validation as the reference's parse_configfile
(/root/reference/miR_PREFeR.py:80-186); errors go to stderr and exit with status -1 like the reference."""
import os
import sys

DEFAULTS = {
    "CONFIG_FILE": "", "FASTA_FILE": "", "ALIGNMENT_FILE": [], "GFF_FILE_EXCLUDE": "", "GFF_FILE_INCLUDE": "",
    "PRECURSOR_LEN": 300, "READS_DEPTH_CUTOFF": 10, "MAX_GAP": 100, "NUM_OF_CORE": 1, "OUTFOLDER": "./", "TMPFOLDER": "",
    "NAME_PREFIX": "", "PIPELINE_PATH": "", "DELETE_IF_SUCCESS": "Y", "CHECKPOINT_SIZE": 3000, "MIN_MATURE_LEN": 18,
    "MAX_MATURE_LEN": 23, "ALLOW_3NT_OVERHANG": False, "ALLOW_NO_STAR_EXPRESSION": True, "OUTPUT_DETAILS_FOR_DEBUG": False,
}


def _die(msg):
    sys.stderr.write(msg)
    sys.exit(-1)


def parse_configfile(configfile):
    if not os.path.exists(os.path.expanduser(configfile)):
        _die("Configuration file " + configfile + " does not exist!!\n")
    opt = {k: (list(v) if isinstance(v, list) else v) for k, v in DEFAULTS.items()}
    opt["CONFIG_FILE"] = configfile
    with open(configfile) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            sp = line.strip().split("=")
            if not (len(sp) > 1 and sp[1]):
                continue
            key, val = sp[0].strip(), sp[1].strip()
            if key == "ALIGNMENT_FILE":
                for name in sp[1].split(","):
                    name = os.path.expanduser(name.strip())
                    if not os.path.exists(name):
                        _die("File " + name + " does not exist!!\n")
                    opt[key].append(name)
            elif key in ("GFF_FILE_EXCLUDE", "GFF_FILE_INCLUDE", "FASTA_FILE"):
                if not os.path.exists(os.path.expanduser(val)):
                    _die("File " + val + " does not exist!!\n")
                opt[key] = os.path.expanduser(val)
            elif key == "NUM_OF_CORE":
                cpucount, n = os.cpu_count() or 1, int(val)
                if 2 * cpucount < n:
                    sys.stderr.write("Warnning: 2*NUM_OF_CORE is larger than CPUS/Cores on the machine. Use " + str(2 * cpucount) + " instead.\n")
                    n = 2 * cpucount
                opt[key] = n
            elif key in ("PRECURSOR_LEN", "MAX_MATURE_LEN", "MIN_MATURE_LEN", "READS_DEPTH_CUTOFF", "MAX_GAP", "CHECKPOINT_SIZE"):
                opt[key] = int(val)
            elif key in ("OUTFOLDER", "TMPFOLDER", "NAME_PREFIX", "DELETE_IF_SUCCESS"):
                opt[key] = val
            elif key in ("ALLOW_NO_STAR_EXPRESSION", "ALLOW_3NT_OVERHANG"):
                v = val.lower()
                if v not in ("y", "n"):
                    _die("Value for ALLOW_3NT_OVERHANG/ALLOW_NO_STAR_EXPRESSION must be one of 'Y/y/N/n'.\n")
                opt[key] = v == "y"
            elif key == "PIPELINE_PATH":
                opt[key] = os.path.expanduser(val)  # the reference checks that its own script lives there; nothing is loaded from it here
    ok = True
    if opt["PRECURSOR_LEN"] < 60 or opt["PRECURSOR_LEN"] > 3000:
        sys.stderr.write("Error: allowed precursor range: 60-3000\n"); ok = False
    if opt["MIN_MATURE_LEN"] > opt["MAX_MATURE_LEN"]:
        sys.stderr.write("Error: MIN_MATURE_LEN is greater than MAX_MATURE_LEN.\n"); ok = False
    if opt["READS_DEPTH_CUTOFF"] < 2:
        sys.stderr.write("Error: READS_DEPTH_CUTOFF should >=2.\n"); ok = False
    if opt["CHECKPOINT_SIZE"] < 10:
        sys.stderr.write("Error: CHECKPOINT_SIZE should >=10.\n"); ok = False
    if opt["GFF_FILE_INCLUDE"] and opt["GFF_FILE_EXCLUDE"]:
        sys.stderr.write("Error: GFF_FILE_EXCLUDE and GFF_FILE_INCLUDE are mutual exclusive, please remove one of them.\n"); ok = False
    if not ok:
        sys.exit(-1)
    return opt
