#!/bin/bash
# Dev tool: vienna-1.8.5 fill / epilogue kernel times for several builds of the library (profiles/tools/ab_time185.py per library).
for l in "$@"; do MIRP_LIB=$PWD/mir-prefer_amd/$l python profiles/tools/ab_time185.py 2>&1 | grep "vienna-1.8.5"; done
