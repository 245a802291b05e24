#!/usr/bin/env python3
"""Dev-only: read the Turner-1999 nearest-neighbour parameter tables (as shipped in ViennaRNA 1.8.5) out of the data segment of the
reference's bundled Linux RNALfold ELF binary and emit them as a plain C header of integer tables (data, not code).

Runs only in the build container (needs /root/reference). Output is committed:
  mir-prefer_amd/csrc/energy_params_t1999.h   (product copy)
  (oracle/lfold185.c includes that file: one copy of the data)
"""
import os, re, struct, subprocess, sys

BIN = "/root/reference/dependency/Linux/x64/RNALfold"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
data = open(BIN, "rb").read()

# program headers (PT_LOAD) for vaddr -> file offset
e_phoff, = struct.unpack_from("<Q", data, 32)
e_phentsize, e_phnum = struct.unpack_from("<HH", data, 54)
LOADS = []
for k in range(e_phnum):
    p_type, p_flags, p_offset, p_vaddr, p_paddr, p_filesz, p_memsz, p_align = struct.unpack_from("<IIQQQQQQ", data, e_phoff + k * e_phentsize)
    if p_type == 1:
        LOADS.append((p_vaddr, p_filesz, p_memsz, p_offset))

def v2o(addr):
    for va, fsz, msz, off in LOADS:
        if va <= addr < va + fsz:
            return off + addr - va
        if va <= addr < va + msz:
            return None           # .bss: zero
    raise KeyError(hex(addr))

syms = {}
for line in subprocess.run([OBJDUMP, "--syms", BIN], capture_output=True, text=True).stdout.splitlines():
    m = re.match(r"^([0-9a-f]{16}) g\s+O \S+\s+([0-9a-f]{16}) (\S+)$", line)
    if m:
        syms[m.group(3)] = (int(m.group(1), 16), int(m.group(2), 16))

def ints(name, n):
    addr, size = syms[name]
    o = v2o(addr)
    if o is None:
        return [0] * n
    return list(struct.unpack_from("<%di" % n, data, o))

def dbl(name):
    return struct.unpack_from("<d", data, v2o(syms[name][0]))[0]

def motif_string(name):
    addr, size = syms[name]
    o = v2o(addr)
    return data[o:o + size].split(b"\0")[0].decode().split()

T = {}
T["stack"] = ints("stack37", 64)
T["hairpin"] = ints("hairpin37", 31)
T["bulge"] = ints("bulge37", 31)
T["internal_loop"] = ints("internal_loop37", 31)
T["mismatchI"] = ints("mismatchI37", 200)
T["mismatchH"] = ints("mismatchH37", 200)
T["dangle5"] = ints("dangle5_37", 40)
T["dangle3"] = ints("dangle3_37", 40)
T["int11"] = ints("int11_37", 8 * 8 * 25)
T["int21"] = ints("int21_37", 8 * 8 * 125)
T["int22"] = ints("int22_37", 8 * 8 * 625)
tetra = motif_string("Tetraloops")
T["Tetraloop_E"] = ints("TETRA_ENERGY37", len(tetra))
tri = motif_string("Triloops")
scal = {"ML_BASE": ints("ML_BASE37", 1)[0], "ML_closing": ints("ML_closing37", 1)[0], "ML_intern": ints("ML_intern37", 1)[0],
        "TerminalAU": ints("TerminalAU", 1)[0], "MAX_NINIO": ints("MAX_NINIO", 1)[0], "ninio": ints("F_ninio37", 5)[2]}
lxc = dbl("lxc37")

def carr(name, vals, dims):
    s = "static const int T99_%s%s = {" % (name, "".join("[%d]" % d for d in dims))
    body = []
    for k in range(0, len(vals), 20):
        body.append("    " + ", ".join(str(v) for v in vals[k:k + 20]) + ",")
    return s + "\n" + "\n".join(body) + "\n};\n"

out = ["// Turner-1999 nearest-neighbour parameters at 37 C as shipped in ViennaRNA 1.8.5 (values read from the reference's bundled",
       "// dependency/Linux/x64/RNALfold by tests/golden/tools/extract_params_t1999.py).  Integer energies in 0.01 kcal/mol.  Data only.",
       "#pragma once", ""]
out.append(carr("stack", T["stack"], [8, 8]))
for nm in ("hairpin", "bulge", "internal_loop"):
    out.append(carr(nm, T[nm], [31]))
for nm in ("mismatchI", "mismatchH"):
    out.append(carr(nm, T[nm], [8, 5, 5]))
for nm in ("dangle5", "dangle3"):
    out.append(carr(nm, T[nm], [8, 5]))
out.append(carr("int11", T["int11"], [8, 8, 5, 5]))
out.append(carr("int21", T["int21"], [8, 8, 5, 5, 5]))
out.append(carr("int22", T["int22"], [8, 8, 5, 5, 5, 5]))
out.append("#define T99_N_TETRALOOPS %d" % len(tetra))
out.append("static const char T99_Tetraloops[%d][8] = {%s};" % (len(tetra), ", ".join('"%s"' % t for t in tetra)))
out.append(carr("Tetraloop_E", T["Tetraloop_E"], [len(tetra)]))
out.append("#define T99_N_TRILOOPS %d" % len(tri))
for k, v in scal.items():
    out.append("#define T99_%s %d" % (k, v))
out.append("#define T99_LXC %r" % lxc)
text = "\n".join(out) + "\n"
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for dst in ("mir-prefer_amd/csrc/energy_params_t1999.h",):          # the one copy: the oracle includes it from there
    open(os.path.join(root, dst), "w").write(text)
print(scal, lxc, len(tetra), tetra[:5], T["Tetraloop_E"][:5], len(tri), T["hairpin"][:10], T["stack"][8:16])
