cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "bge 64" "bge 256"; do
  set -- $cfg
  python3 tools/enc_chain_profile.py $1 $2
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chain2_$1_$2 -- python3 tools/enc_chain_profile.py $1 $2 > /dev/null 2>&1
  f=$(ls -t gpurun_out/chain2_$1_$2/*/*kernel_stats.csv | head -1)
  python3 - <<PY
import csv
rows=list(csv.reader(open("$f")))
for r in rows[1:12]:
    if "crs" in r[0]: print("   ", r[0][22:80].ljust(60), r[1].rjust(6), ("%.1f"%(float(r[3])/1000)).rjust(8), "us  min", ("%.1f"%(float(r[5])/1000)))
PY
done
