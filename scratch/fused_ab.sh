#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --no-roofline --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
run OMR_FUSED_BWD=15
run OMR_FUSED_BWD=7
run OMR_FUSED_BWD=15
run OMR_FUSED_BWD=7
run OMR_FUSED_BWD=15
run OMR_FUSED_BWD=7
