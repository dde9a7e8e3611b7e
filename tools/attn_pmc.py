"""profiles/rNN_attention_pmc_SQ.json from a `rocprofv3 --pmc ... -- python3 tools/attn_shapes.py` run:
python tools/attn_pmc.py <dir> <out.json>.  Mean per launch of every collected counter for the three attention kernels, plus
derived ratios (vector instructions per MFMA, share of wave time issuing / waiting)."""
import csv, glob, json, sys
d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
fam = {"fwd": "attn_fwd_kernel", "dq": "attn_bwd_dq_kernel", "dkv": "attn_bwd_dkv_kernel", "words": "attn_dropout_words_kernel"}
acc = {k: {} for k in fam}
for r in csv.DictReader(open(f)):
    for k, pat in fam.items():
        if pat in r["Kernel_Name"]:
            a = acc[k].setdefault(r["Counter_Name"], {})
            a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
res = {}
for k, cs in acc.items():
    if not cs:
        continue
    m = {c: sum(v.values()) / len(v) for c, v in cs.items()}
    mf = m.get("SQ_INSTS_VALU_MFMA_BF16") or m.get("SQ_INSTS_MFMA") or 0.0
    if mf:
        m["valu_per_mfma"] = round((m["SQ_INSTS_VALU"] - mf) / mf, 2)
    if m.get("SQ_WAVE_CYCLES"):
        for c, name in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_ANY", "waiting"), ("SQ_WAIT_INST_ANY", "issue_stalled")):
            if c in m:
                m["frac_" + name] = round(m[c] / m["SQ_WAVE_CYCLES"], 3)
    m["launches"] = len(next(iter(cs.values())))
    res[k] = m
json.dump({"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_BF16 SQ_ACTIVE_INST_LDS -- python3 tools/attn_shapes.py",
           "shape": "cross-attention at C2: B=32, 4 heads x 64, T=512 queries, S=4096 keys, bf16; mean per launch over the four (dropout, key bias) combinations of the tool",
           "note": "SQ_INSTS_VALU counts every vector instruction including the MFMAs (valu_per_mfma subtracts them); SQ_* cycle counters are quad-cycles summed over waves",
           "kernels": res}, open(out, "w"), indent=1)
print(json.dumps({k: {c: v for c, v in m.items() if c.startswith(("valu_per", "frac_", "launches"))} for k, m in res.items()}, indent=1))
