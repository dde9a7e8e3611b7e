#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun; outputs under gpurun_out/r3p/)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 bench.py --no-roofline --no-cpu-baseline --steps 10 > $O/train.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_noside -- python3 bench.py --no-roofline --no-cpu-baseline --no-side-stream --steps 10 > $O/train_noside.log 2>&1 || exit 1
python tools/prof_summary.py --families $O/train > $O/families_shipped.txt
python tools/prof_summary.py --families $O/train_noside > $O/families_noside.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dom -- python3 tools/run_dominant_kernel.py > $O/dom.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dom_fetch -- python3 tools/run_dominant_kernel.py > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dom_write -- python3 tools/run_dominant_kernel.py > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_BF16 SQ_ACTIVE_INST_LDS --output-format csv -d $O/attn_pmc -- python3 tools/attn_shapes.py > $O/attn_pmc.log 2>&1 || exit 1
python tools/attn_pmc.py $O/attn_pmc $O/attention_pmc_SQ.json > $O/attention_pmc_summary.txt
for c in c1 c3 c3mel c4; do python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> /dev/null || exit 1; done
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err || exit 1
python tools/host_overhead.py > $O/host.txt 2>&1
echo done
