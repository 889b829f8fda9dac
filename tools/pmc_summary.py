#!/usr/bin/env python3
"""gpurun_out/<TAG>_* (tools/make_profiles.sh) -> profiles/<TAG>_{bench_*.json, *_kernel_stats.csv, pmc_summary.json}.
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE is reported in KiB and, on gfx950, at HALF the
bytes of a wide coalesced streaming read -> read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE is reported raw."""
import csv, glob, json, os, re, shutil, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(pr, exist_ok=True)
csv.field_size_limit(1 << 30)
for w in ("c2", "c3", "c4", "c5", "enc-minilm", "enc-bge"):
    src = os.path.join(go, f"{tag}_bench_{w}.json")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(pr, f"{tag}_bench_{w}.json"))
for w in ("c2", "c3", "c4", "c5", "enc-minilm", "enc-bge"):
    found = sorted(glob.glob(os.path.join(go, f"{tag}_stats_{w}", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    for f in found[-1:]:   # gpurun merges every call's files into gpurun_out/: keep the newest run only
        rows = list(csv.reader(open(f)))
        with open(os.path.join(pr, f"{tag}_{w.replace('-', '_')}_kernel_stats.csv"), "w", newline="") as fh:
            wr = csv.writer(fh)
            for r in rows[:26]:
                r[0] = r[0][:160]
                wr.writerow(r)
scan_pat = re.compile(r"crs::.*?(scan_(?:tb|i8|wide|w2|f16|f16_ring)_kernel<[^>]*>)")   # ours only (rocprim has *scan* kernels too)
summary = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace --output-format csv -- python3 bench.py "
           "--workload <w> --steps 6 --warmup 1 --streams 1 --no-graph --no-cpu-baseline (tools/make_profiles.sh); per-launch means over the scan "
           "kernel's dispatches (first two skipped); hbm_read_bytes = FETCH_SIZE(KiB) * 1024 * 2 (gfx950: FETCH_SIZE reports half of a wide "
           "coalesced stream, MI355X_MICROARCH.md section HBM); WRITE_SIZE is uncalibrated for this kernel's scattered 4-byte stores and is "
           "reported raw. Under --pmc the profiler idles the GPU between dispatches, so kernel_ns_under_pmc is an isolated-launch duration."}
for w in ("c4",):
    ent = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals, ns, name, vg = [], [], None, None
        found = sorted(glob.glob(os.path.join(go, f"{tag}_pmc_{c}_{w}", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        for f in found[-1:]:
            for r in csv.DictReader(open(f)):
                m = scan_pat.search(r["Kernel_Name"])
                if not m or r["Counter_Name"] != c: continue
                name = m.group(1); vg = int(r["VGPR_Count"])
                vals.append(float(r["Counter_Value"])); ns.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        if len(vals) > 4: vals, ns = vals[2:], ns[2:]
        if vals:
            ent["kernel"] = name; ent["vgpr"] = vg; ent["dispatches"] = len(vals)
            ent[f"{c}_KiB"] = sum(vals) / len(vals)
            ent["kernel_ns_under_pmc"] = sum(ns) / len(ns)
    if "FETCH_SIZE_KiB" in ent:
        ent["hbm_read_bytes_per_launch"] = int(ent["FETCH_SIZE_KiB"] * 1024 * 2)
        bj = os.path.join(go, f"{tag}_bench_{w}.json")
        if os.path.exists(bj):
            try:
                bl = json.loads(open(bj).read().strip().splitlines()[-1])
                ent["rows"] = bl["config"]["rows_per_gpu"]; ent["algorithmic_bytes"] = bl["roofline"]["algorithmic_bytes"]
                ent["traffic_over_algorithmic"] = round(ent["hbm_read_bytes_per_launch"] / ent["algorithmic_bytes"], 4)
            except Exception:
                pass
        summary[f"{w}-n1"] = ent
json.dump(summary, open(os.path.join(pr, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:1500])
