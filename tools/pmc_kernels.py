"""Per-kernel means of the counters of one or more `rocprofv3 --pmc ... --output-format csv -d DIR -- python3 <cmd>` passes:
python tools/pmc_kernels.py DIR [DIR ...] [--min-us N].  Kernels are keyed by (name, grid, workgroup) so the instantiations a
demangler collapses stay apart; derived columns: share of wave time issuing / parked at s_waitcnt or a barrier / issue-stalled,
vector instructions per MFMA, MFMA-busy share of the CU-cycles, LDS-active share."""
import csv, glob, sys, collections

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
acc = collections.defaultdict(lambda: collections.defaultdict(dict))
for d in dirs:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"][:96], r["Grid_Size"], r["Workgroup_Size"], r.get("LDS_Block_Size", ""))
            a = acc[key][r["Counter_Name"]]
            a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
rows = []
for key, cs in acc.items():
    m = {c: sum(v.values()) / len(v) for c, v in cs.items()}
    m["n"] = max(len(v) for v in cs.values())
    rows.append((key, m))
rows.sort(key=lambda km: -km[1].get("SQ_WAVE_CYCLES", 0) * km[1]["n"])
hdr = f"{'kernel':70s} {'grid':>8s} {'n':>4s} {'issue':>6s} {'wait':>6s} {'stall':>6s} {'valu/mfma':>9s} {'valu_act':>8s} {'lds_act':>7s} {'mfma_busy':>9s} {'busy_cyc':>10s}"
print(hdr)
for key, m in rows:
    wc = m.get("SQ_WAVE_CYCLES")
    if not wc:
        continue
    fr = lambda c: f"{m[c] / wc:6.3f}" if c in m else "     -"
    mf = m.get("SQ_INSTS_VALU_MFMA_BF16") or m.get("SQ_INSTS_MFMA") or 0
    vpm = f"{(m['SQ_INSTS_VALU'] - mf) / mf:9.1f}" if mf and "SQ_INSTS_VALU" in m else "        -"
    busy = m.get("SQ_BUSY_CYCLES") or m.get("GRBM_GUI_ACTIVE")
    mb = f"{m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES']:9.3f}" if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("SQ_BUSY_CYCLES") else "        -"
    print(f"{key[0][:70]:70s} {key[1]:>8s} {m['n']:4d} {fr('SQ_ACTIVE_INST_ANY')} {fr('SQ_WAIT_ANY')} {fr('SQ_WAIT_INST_ANY')} {vpm} {fr('SQ_ACTIVE_INST_VALU'):>8s} {fr('SQ_ACTIVE_INST_LDS'):>7s} {mb} {busy or 0:10.0f}")
