"""Raw per-kernel means of every counter of a rocprofv3 --pmc pass: python tools/pmc_raw.py DIR [substr]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:90], r["Grid_Size"])
        a = acc[key][r["Counter_Name"]]
        a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for key, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", {0: 0}).values())):
    m = {c: sum(v.values()) / len(v) for c, v in cs.items()}
    print(key[0], key[1], "n=%d" % max(len(v) for v in cs.values()))
    print("   ", "  ".join(f"{c}={v:.4g}" for c, v in sorted(m.items())))
    if m.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print(f"    mfma_busy/busy={m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES']:.3f}", end="")
    if m.get("SQ_LDS_IDX_ACTIVE"):
        print(f"  lds_conflict/lds_active={m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.3f}  lds_active/busy={m['SQ_LDS_IDX_ACTIVE'] / max(m.get('SQ_BUSY_CYCLES', 1), 1):.3f}", end="")
    print()
