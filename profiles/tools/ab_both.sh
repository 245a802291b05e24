for l in "$@"; do MIRP_LIB=$PWD/mir-prefer_amd/$l python profiles/tools/ab_time.py 2>&1 | tail -1; MIRP_LIB=$PWD/mir-prefer_amd/$l python profiles/tools/ab_time185.py 2>&1 | tail -2 | head -1; done
