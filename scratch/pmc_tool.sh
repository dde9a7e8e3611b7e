#!/bin/bash
# usage: bash scratch/pmc_tool.sh <outdir-name> <python tool + args...>   (two SQ passes, summaries printed)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_BF16 SQ_ACTIVE_INST_LDS --output-format csv -d $O/p1 -- python3 "$@" > $O/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/p2 -- python3 "$@" > $O/p2.log 2>&1 || exit 1
python tools/pmc_kernels.py $O/p1 > $O/p1.txt
python tools/pmc_raw.py $O/p2 > $O/p2.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 "$@" > $O/tr.log 2>&1 || exit 1
echo done
