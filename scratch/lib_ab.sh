#!/bin/bash
# in-one-call A/B of two builds of the library: scratch/libomr_hip_base.so (a copy of the build to compare against) vs the in-tree one
# usage: lib_ab.sh [extra bench.py flags]
run() { echo "== $*"; env "$1" python bench.py --no-cpu-baseline --no-roofline --steps 30 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
run OMR_X=new "$@"
run OMR_HIP_LIB=$PWD/scratch/libomr_hip_base.so "$@"
run OMR_X=new "$@"
run OMR_HIP_LIB=$PWD/scratch/libomr_hip_base.so "$@"
