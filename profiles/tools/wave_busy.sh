#!/bin/bash
# Dev tool: per-wave busy / barrier-wait ticks of the fill kernel on the bench workload, product code + two clock reads per wave and interval
# (make VARIANT=<name> VFLAGS="-DMIRP_LITE_CLOCKS ..."):   profiles/tools/wave_busy.sh libmirprefer_vlite.so [more libraries]
for l in "$@"; do
  echo "== $l"
  MIRP_LIB=$PWD/mir-prefer_amd/$l MIRP_FOLD_CLOCKS=2 python profiles/tools/ab_time.py 2>&1 | grep -E "wave +[0-9]+:|fill " | tail -17 | \
    awk '/wave/ { split($0, a, "splits="); split(a[2], b, " barrier="); printf "%s busy %.1f wait %.1f (%.0f %% busy)\n", $5, b[1] / 1e9, b[2] / 1e9, 100 * b[1] / (b[1] + b[2]); next } { print }'
done
