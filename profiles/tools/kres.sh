#!/bin/bash
# Dev tool: register / scratch / LDS use of the kernels of one source file (device-only compile with the product flags + extra flags):
#   profiles/tools/kres.sh fold_lds_kernel.hip [name filter] [extra flags]      -> also leaves the disassembly in /tmp/kres_<file>.s
cd "$(dirname "$0")/../../mir-prefer_amd/csrc" || exit 1
src=$1; f=${2:-.}; shift; shift
out=/tmp/kres_$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-missing-braces --cuda-device-only -c "$@" $src -o $out.co || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$out.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$out.elf && mv $out.elf $out.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out.co | grep -E "^ *\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):" | paste - - - - - - | sed 's/  */ /g' | grep -E "$f"
/opt/rocm/lib/llvm/bin/llvm-objdump -d $out.co > $out.s
