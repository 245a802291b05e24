#!/bin/bash
# Dev tool: coverage stage time with the fused scan (default) and with the atomic scatter path (MIRP_COV_FUSED=0; the tool pins the path with mirp_set_coverage_path), profiles/tools/cov_time.py
MIRP_COV_FUSED=1 timeout 200 python profiles/tools/cov_time.py 2>&1 | grep -E "coverage"
MIRP_COV_FUSED=0 timeout 200 python profiles/tools/cov_time.py 2>&1 | grep -E "coverage"
