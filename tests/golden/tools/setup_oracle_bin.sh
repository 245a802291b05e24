#!/bin/sh
# Dev-only (build container): wrappers that make the reference's bundled third-party binaries runnable
# here, for golden-vector generation. Nothing here travels to the GPU box.
set -e
B=${MIRP_ORACLE_BIN:-/tmp/ora/bin}; L=$(dirname "$B")/lib
mkdir -p "$B" "$L"
gcc -O1 -o "$B/mloader" "$(dirname "$0")/mloader.c" -ldl
ln -sf /usr/lib/x86_64-linux-gnu/libncurses.so.6 "$L/libncurses.so.5"
ln -sf /usr/lib/x86_64-linux-gnu/libtinfo.so.6 "$L/libtinfo.so.5"
printf '#!/bin/sh\nLD_LIBRARY_PATH=%s exec /root/reference/dependency/Linux/x64/samtools "$@"\n' "$L" > "$B/samtools"
printf '#!/bin/sh\nexec %s/mloader /root/reference/dependency/Mac/osx-10.9/RNALfold-2.1.2 "$@"\n' "$B" > "$B/RNALfold212"
ln -sf /root/reference/dependency/Linux/x64/RNALfold "$B/RNALfold185"
chmod +x "$B/samtools" "$B/RNALfold212"
