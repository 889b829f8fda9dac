#!/usr/bin/env python3
"""gpurun_out/r03_* (tools/make_profiles_r03.sh) -> profiles/r03_*: bench lines, kernel-stats tables, PMC / SQ summaries.
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KiB and, on gfx950, HALF the bytes of a wide coalesced streaming
read -> read bytes = FETCH_SIZE * 1024 * 2."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# usage: collect_profiles_r03.py [OUT_DIR]   (default profiles/; on the GPU box: gpurun_out/r03_profiles, the raw traces are then deleted)
go, pr, tag = os.path.join(root, "gpurun_out"), (os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(root, "profiles")), "r03"
os.makedirs(pr, exist_ok=True)
csv.field_size_limit(1 << 30)
for f in glob.glob(os.path.join(go, f"{tag}_bench_*.json")):
    if os.path.getsize(f):
        shutil.copy(f, os.path.join(pr, os.path.basename(f)))
for f in glob.glob(os.path.join(go, f"{tag}_sq_*.json")) + glob.glob(os.path.join(go, f"{tag}_gemm_tflops.txt")):
    shutil.copy(f, os.path.join(pr, os.path.basename(f).replace("_sq_", "_sq_counters_")))
for w in ("c2", "c3", "c4", "c5", "enc-minilm", "enc-bge"):
    found = sorted(glob.glob(os.path.join(go, f"{tag}_stats_{w}", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    for f in found[-1:]:
        rows = list(csv.reader(open(f)))
        names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows[1:26]), capture_output=True, text=True).stdout.splitlines()
        with open(os.path.join(pr, f"{tag}_{w.replace('-', '_')}_kernel_stats.csv"), "w", newline="") as fh:
            wr = csv.writer(fh)
            wr.writerow(rows[0])
            for r, n in zip(rows[1:26], names):
                r[0] = n[:160]
                wr.writerow(r)
scan_pat = re.compile(r"crs::.*?(scan_(?:tb|i8|wide|w2|f16|f16_ring)_kernel<[^>]*>)")
summary = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace --output-format csv -- python3 bench.py --workload c4 "
           "--steps 6 --warmup 1 --streams 1 --no-graph --no-cpu-baseline --recall-queries 64 (tools/make_profiles_r03.sh); per-launch means over the "
           "scan kernel's dispatches (first two skipped); hbm_read_bytes = FETCH_SIZE(KiB) * 1024 * 2 (gfx950: FETCH_SIZE reports half of a wide coalesced "
           "stream, MI355X_MICROARCH.md section HBM); WRITE_SIZE raw (uncalibrated for this kernel's scattered 4-byte stores)."}
ent = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals, ns, name = [], [], None
    found = sorted(glob.glob(os.path.join(go, f"{tag}_pmc_{c}_c4", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in found[-1:]:
        for r in csv.DictReader(open(f)):
            m = scan_pat.search(r["Kernel_Name"])
            if not m or r["Counter_Name"] != c or re.search(r", ?10>$", m.group(1)): continue     # the timed plan, not the k' = 10 comparison launch
            name = m.group(1)
            vals.append(float(r["Counter_Value"])); ns.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if len(vals) > 4: vals, ns = vals[2:], ns[2:]
    if vals:
        ent.update(kernel=name, dispatches=len(vals))
        ent[f"{c}_KiB"] = sum(vals) / len(vals)
        ent["kernel_ns_under_pmc"] = sum(ns) / len(ns)
if "FETCH_SIZE_KiB" in ent:
    ent["hbm_read_bytes_per_launch"] = int(ent["FETCH_SIZE_KiB"] * 1024 * 2)
    try:
        bl = json.loads(open(os.path.join(go, f"{tag}_bench_c4.json")).read().strip().splitlines()[-1])
        ent["rows"] = bl["config"]["rows_per_gpu"]; ent["algorithmic_bytes"] = bl["roofline"]["algorithmic_bytes"]
        ent["traffic_over_algorithmic"] = round(ent["hbm_read_bytes_per_launch"] / ent["algorithmic_bytes"], 4)
    except Exception:
        pass
    summary["c4-n1"] = ent
json.dump(summary, open(os.path.join(pr, f"{tag}_pmc_summary.json"), "w"), indent=1)
# GEMM SQ counters per kernel x shape (dispatch order = shape order of the bench_gemm command; 14 launches per shape)
gem = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    found = sorted(glob.glob(os.path.join(go, f"{tag}_gemm_sq{i}", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in found[-1:]:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm" not in k or "crs" not in k: continue
            key = (re.sub(r".*?(gemm\w*_kernel)(I[^E]*E)?.*", r"\1\2", k), r["Grid_Size"], r["LDS_Block_Size"])
            gem[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            gem[key].setdefault("_ns", []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {"_how": "rocprofv3 --pmc <SQ set> --kernel-trace -- python3 tools/bench_gemm.py <shapes> (tools/make_profiles_r03.sh); per-dispatch means, keyed by (kernel, "
               "grid size, LDS): mfma_busy_per_simd = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES); clock_ghz = SQ_BUSY_CU_CYCLES / 256 CUs / duration; "
               "TFLOP/s per shape unprofiled: r03_gemm_tflops.txt"}
for key, d in gem.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = {c: round(v) for c, v in m.items() if not c.startswith("_")}
    e["dispatches"] = len(d["_ns"]); e["kernel_us_under_pmc"] = round(m["_ns"] / 1e3, 1)
    if m.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_busy_per_simd"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * m["SQ_BUSY_CU_CYCLES"]), 3)
        e["clock_ghz"] = round(m["SQ_BUSY_CU_CYCLES"] / 256 / m["_ns"], 3)
    if m.get("SQ_WAVE_CYCLES"):
        e["wait_any"] = round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 3)
    out[" | ".join(key)] = e
json.dump(out, open(os.path.join(pr, f"{tag}_gemm_sq_counters.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:1200])
print(sorted(os.listdir(pr))[-40:])
