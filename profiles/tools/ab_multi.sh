#!/bin/bash
# Dev tool: fill / epilogue kernel times of the bench workload for several builds of the library on one box (profiles/tools/ab_time.py per library).
# usage: profiles/tools/ab_multi.sh libmirprefer.so libmirprefer_vX.so ...
for l in "$@"; do MIRP_LIB=$PWD/mir-prefer_amd/$l python profiles/tools/ab_time.py 2>&1 | tail -${TAILN:-1}; done
