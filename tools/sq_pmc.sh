#!/bin/bash
# SQ counter passes (matrix pipe busy, LDS activity / conflicts, wave wait cycles, in-kernel clock) for one bench workload's scan
# kernel.  Counters only with --kernel-trace, one small set per run.   usage: tools/sq_pmc.sh TAG WORKLOAD
# The scan's dispatches are split by what ran just before them: "back_to_back" (the previous dispatch was the same scan kernel:
# bench.py's crs_time_cosine_topk loop, the figure roofline.kernel_ms carries) and "in_mix" (anything else ran in between: the step
# loop, where encoder / merge / refine kernels separate two scans) -- the profiler serialises dispatches, so this separates the
# THERMAL / CLOCK state a launch starts in, not concurrency.
set -o pipefail
TAG=${1:-r03}; W=${2:-c5}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq${i}_$W -- python3 bench.py --workload $W --steps 6 --warmup 1 --streams 1 --no-graph --no-cpu-baseline --recall-queries 64 > gpurun_out/${TAG}_sq${i}_$W.log 2>&1 || { tail -5 gpurun_out/${TAG}_sq${i}_$W.log; exit 1; }
done
python3 - $TAG $W <<'PY'
import csv, glob, json, os, sys, collections
tag, w = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    fs = sorted(glob.glob(f"gpurun_out/{tag}_sq{i}_{w}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(fs[-1])))
    # one row per (dispatch, counter): order dispatches by start time, remember each dispatch's predecessor kernel
    disp = {}
    for r in rows:
        disp.setdefault(r["Dispatch_Id"], (float(r["Start_Timestamp"]), r["Kernel_Name"]))
    order = sorted(disp, key=lambda d: disp[d][0])
    prev = {d: (disp[order[j - 1]][1] if j else "") for j, d in enumerate(order)}
    for r in rows:
        k = r["Kernel_Name"]
        if "crs::" not in k or "scan_" not in k: continue
        name = k[k.index("scan_"):][:60]
        grp = "back_to_back" if prev[r["Dispatch_Id"]] == k else "in_mix"
        acc[(name, grp)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[(name, grp)].setdefault("_ns", []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {"_how": f"rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --workload {w} --steps 6 --warmup 1 --streams 1 --no-graph "
               "--no-cpu-baseline --recall-queries 64 (tools/sq_pmc.sh; two passes of six SQ counters); per-dispatch means over the scan kernel's "
               "launches, split by the dispatch that preceded them (back_to_back: the same scan kernel, i.e. the crs_time_cosine_topk loop; in_mix: "
               "another kernel, i.e. the step loop). Ratios: mfma_busy_per_simd = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES); lds_active = "
               "SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES; wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES; clock_ghz = SQ_BUSY_CU_CYCLES / 256 CUs / duration "
               "(serialised launch under the profiler)."}
for (name, grp), d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = {c: round(v) for c, v in m.items() if not c.startswith("_")}
    e["dispatches"] = len(d["_ns"]) // 6
    e["kernel_us_under_pmc"] = round(m["_ns"] / 1e3, 1)
    if m.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_busy_per_simd"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * m["SQ_BUSY_CU_CYCLES"]), 3)
        e["lds_active"] = round(m.get("SQ_LDS_IDX_ACTIVE", 0) / m["SQ_BUSY_CU_CYCLES"], 3)
        e["clock_ghz"] = round(m["SQ_BUSY_CU_CYCLES"] / 256 / m["_ns"], 3)
    if m.get("SQ_WAVE_CYCLES"):
        e["wait_any"] = round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 3)
    out[f"{name} [{grp}]"] = e
json.dump(out, open(f"gpurun_out/{tag}_sq_{w}.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])
PY
