#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (scripts/pmc_round.sh): per kernel, the mean of every counter per dispatch plus a few ratios."""
import csv, glob, json, os, re, sys
root = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "dgs::" not in name:
            continue
        short = re.sub(r"\(.*", "", name).replace("void ", "").strip()
        key = (short, r["Counter_Name"], r["Dispatch_Id"])
        per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    for (k, c, _), v in per.items():
        a = acc.setdefault(k, {}).setdefault(c, [0.0, 0])
        a[0] += v
        a[1] += 1
out = {}
for k, cs in sorted(acc.items()):
    d = {c: v[0] / v[1] for c, v in cs.items()}
    d["dispatches"] = max(v[1] for v in cs.values())
    g = d.get
    if g("SQ_WAVE_CYCLES"):
        d["valu_active_per_wave_cycle"] = g("SQ_ACTIVE_INST_VALU", 0) / g("SQ_WAVE_CYCLES")
        d["any_inst_active_per_wave_cycle"] = g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
    if g("SQ_BUSY_CYCLES") and g("SQ_WAVE_CYCLES"):
        d["resident_waves_per_busy_cycle"] = g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CYCLES")
    if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
        d["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
        d["salu_insts_per_wave"] = g("SQ_INSTS_SALU", 0) / g("SQ_WAVES")
        d["vmem_rd_insts_per_wave"] = g("SQ_INSTS_VMEM_RD", 0) / g("SQ_WAVES")
    if g("TCC_REQ_sum") and g("TCC_HIT_sum") is not None:
        d["l2_hit_rate"] = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum", 0), 1)
    out[k] = d
json.dump(out, sys.stdout, indent=1, sort_keys=True)
