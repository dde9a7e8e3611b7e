"""Summaries of a rocprofv3 run of bench.py (CSV output).

    python tools/prof_summary.py <dir> [steps+warmup] [rows]          per-kernel table from *_kernel_stats.csv (--stats run)
    python tools/prof_summary.py --families <dir> [--steps K]         per-family ms/step table from *_kernel_trace.csv

--families reads the per-dispatch trace, cuts it into training steps at the Adam launches (one per step in the unimodal
configurations), keeps the last K steps (default: all but the first three = the warm-up) and prints, per kernel family, launches
per step, summed kernel time per step and the share on the busiest queue (the main stream); plus the busy time (union of kernel
intervals) per queue and overall.  Run it on a `bench.py --no-roofline --no-cpu-baseline` trace: that process contains only
the training steps (no decode benchmark, no roofline micro-benchmark)."""
import csv
import glob
import os
import re
import sys

FAMILIES = [      # first match wins
    ("conv backward (one pass: dgrad + wgrad [+ norm apply])", r"conv_bwd_fused"),
    ("conv wgrad", r"wgrad_dma|conv3x3_wgrad|conv1_wgrad"),
    ("depthwise", r"dwconv"),
    ("conv fwd/dgrad", r"conv3x3_mfma|conv1_direct|conv3x3_"),
    ("attention", r"attn_"),
    ("gemm dW (grouped)", r"gemm_dw_grouped"),
    ("gemm", r"gemm"),
    ("instancenorm", r"instnorm"),
    ("layernorm", r"add_ln|layernorm|ln_"),
    ("cross-entropy", r"ce_"),
    ("adam", r"adam"),
    ("weight flip", r"weight_flip"),
    ("dropout mask gen", r"attn_mask_gen|dropmask"),
    ("elementwise (ours)", r"embed|add_pe2d|relu_bwd|dropout|colsum|cast_kernel|add_kernel|argmax|topk|memset|fill"),
    ("torch elementwise", r"at::native|at_cuda|elementwise_kernel|vectorized"),
]


def short(name):
    return name.replace("_ZN12_GLOBAL__N_1", "").replace("(anonymous namespace)::", "")


def family(name):
    for fam, pat in FAMILIES:
        if re.search(pat, name):
            return fam
    return "other"


def union_ms(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot / 1e6


def families(d, keep):
    f = max(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    recs = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0")) for r in rows))
    adam = [i for i, r in enumerate(recs) if re.search(r"adam", r[2])]
    if len(adam) < 2:
        sys.exit("no Adam launches in the trace: is this a bench.py training run?")
    if keep <= 0:
        keep = max(1, len(adam) - 1 - 3)
    keep = min(keep, len(adam) - 1)
    lo, hi = adam[-keep - 1], adam[-1]            # (Adam of step n-keep-1, Adam of the last step]
    sel = recs[lo + 1:hi + 1]
    span = (sel[-1][1] - recs[lo][1]) / 1e6
    queues = {}
    for s, e, n, q in sel:
        queues.setdefault(q, []).append((s, e))
    main_q = max(queues, key=lambda q: union_ms(queues[q]))
    fam = {}
    for s, e, n, q in sel:
        a = fam.setdefault(family(n), [0, 0.0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e6
        if q == main_q:
            a[2] += (e - s) / 1e6
    print(f"trace {os.path.relpath(f)}: {keep} steps between Adam launches, {span / keep:.2f} ms/step wall")
    print(f"{'family':24s} {'launches/step':>13s} {'kernel ms/step':>15s} {'on main queue':>14s}")
    for k, (n, t, tm) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:24s} {n / keep:13.1f} {t / keep:15.3f} {tm / keep:14.3f}")
    print(f"{'sum':24s} {sum(a[0] for a in fam.values()) / keep:13.1f} {sum(a[1] for a in fam.values()) / keep:15.3f} {sum(a[2] for a in fam.values()) / keep:14.3f}")
    for q, iv in sorted(queues.items(), key=lambda kv: -union_ms(kv[1])):
        print(f"queue {q}: busy {union_ms(iv) / keep:.3f} ms/step ({len(iv) / keep:.1f} launches/step)" + ("   <- main stream" if q == main_q else ""))
    print(f"all queues: busy {union_ms([(s, e) for s, e, _, _ in sel]) / keep:.3f} ms/step")
    top = {}
    for s, e, n, q in sel:
        a = top.setdefault(n, [0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e6
    print("top kernels (full names as rocprofv3 prints them):")
    for n, (c, t) in sorted(top.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {n[:110]:110s} n/step={c / keep:5.1f} ms/step={t / keep:6.3f} avg={1e3 * t / c:7.1f}us")


def per_kernel(argv):
    d = argv[0]; steps = int(argv[1]) if len(argv) > 1 else 4
    f = max(glob.glob(d + "/*/*_kernel_stats.csv"), key=os.path.getmtime)      # newest run in the directory
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total ms/step", round(tot / 1e6 / steps, 2))
    for r in rows[:int(argv[2]) if len(argv) > 2 else 16]:
        n = short(r["Name"])
        print(f"{n[:84]:84s} n/step={int(r['Calls']) / steps:6.1f} ms/step={float(r['TotalDurationNs']) / steps / 1e6:7.2f} avg={float(r['AverageNs']) / 1e3:8.1f}us {float(r['Percentage']):5.1f}%")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--families":
        keep = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 0
        families(sys.argv[2], keep)
    else:
        per_kernel(sys.argv[1:])
