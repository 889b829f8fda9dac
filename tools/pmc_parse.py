#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel mean of each counter (and duration).
usage: tools/pmc_parse.py DIR [kernel-substring]   (skips the first two dispatches of each kernel)"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "crs::"
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if pat not in name: continue
        m = re.search(r"(\w+_kernel\w*(<[^>]*>)?)", name); short = m.group(1) if m else name[:70]
        key = (r["Dispatch_Id"],)
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[short]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        acc[short]["_vgpr"].append(float(r["VGPR_Count"])); acc[short]["_agpr"].append(float(r["Accum_VGPR_Count"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[2:] if len(v) > 4 else v
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
