"""Per-launch average of one rocprofv3 --pmc counter for the kernels whose name contains a pattern:
python tools/pmc_summary.py <dir> <COUNTER> <pattern>  (reads <dir>/*/*_counter_collection.csv)."""
import csv, glob, sys
d, counter, pat = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
vals = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == counter and pat in r["Kernel_Name"]:
        vals.setdefault(r["Dispatch_Id"], 0.0)
        vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = list(vals.values())
print(counter, "launches", len(v), "mean", sum(v) / max(1, len(v)), "min", min(v), "max", max(v))
