#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 bench.py --no-roofline --no-cpu-baseline --steps 10 > $O/train.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_noside -- python3 bench.py --no-roofline --no-cpu-baseline --no-side-stream --steps 10 > $O/train_noside.log 2>&1 || exit 1
python tools/prof_summary.py --families $O/train > $O/families_shipped.txt
python tools/prof_summary.py --families $O/train_noside > $O/families_noside.txt
echo done
