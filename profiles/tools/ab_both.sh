#!/bin/bash
# usage (GPU box, repo root): bash profiles/tools/ab_both.sh libmirprefer.so libmirprefer_v<variant>.so ...   -- fill-kernel time of each library under both fold models
for l in "$@"; do MIRP_LIB=$PWD/mir-prefer_amd/$l python profiles/tools/ab_time.py 2>&1 | tail -1; MIRP_LIB=$PWD/mir-prefer_amd/$l python profiles/tools/ab_time185.py 2>&1 | tail -2 | head -1; done
