#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/train -- python3 bench.py --no-roofline --no-cpu-baseline --steps 8 > $O/train.log 2>&1 || exit 1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3t/train/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
adam=[i for i,r in enumerate(rows) if 'adam' in r['Kernel_Name']]
for k in (4,5,6):
    lo,hi=adam[k],adam[k+1]
    step=rows[lo:hi+1]
    t0=int(step[0]['Start_Timestamp'])
    mainq=max(set(r['Queue_Id'] for r in step), key=lambda q: sum(1 for r in step if r['Queue_Id']==q))
    m=[r for r in step if r['Queue_Id']==mainq][:-1]
    side=[r for r in step if r['Queue_Id']!=mainq]
    print('step %.2f ms; main last end %.2f; side last end %.2f; side tail kernels:'%((int(step[-1]['End_Timestamp'])-t0)/1e6,(int(m[-1]['End_Timestamp'])-t0)/1e6,(max(int(r['End_Timestamp']) for r in side)-t0)/1e6))
    for r in side[-4:]: print('    %6.2f -> %6.2f  %s'%((int(r['Start_Timestamp'])-t0)/1e6,(int(r['End_Timestamp'])-t0)/1e6,r['Kernel_Name'][:70]))
PY
