"""Per-kernel time summary of a rocprofv3 --kernel-trace --stats --output-format csv run: python tools/prof_summary.py <dir> <steps+warmup> <rows>."""
import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
import os
f = max(glob.glob(d + '/*/*_kernel_stats.csv'), key=os.path.getmtime)      # newest run in the directory
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total ms/step", round(tot/1e6/steps, 2))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    n = r['Name'].replace('_ZN12_GLOBAL__N_1','').replace('(anonymous namespace)::','')
    print(f"{n[:84]:84s} n/step={int(r['Calls'])/steps:6.1f} ms/step={float(r['TotalDurationNs'])/steps/1e6:7.2f} avg={float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.1f}%")
