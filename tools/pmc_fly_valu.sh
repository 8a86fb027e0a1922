#!/bin/bash
# SQ_INSTS_VALU / SQ_WAVES / duration of the one-launch layer on config 2 for a list of plans:
#   PLANS="1,56,16,2,-1,-1,-1,4 1,56,16,4,-1,0,0,4" bash tools/pmc_fly_valu.sh
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/pmc_valu"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
for P in $PLANS; do
  PLAN=$P ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/$P" -o p -- python "$R/tools/run_fly.py" > "$OUT/$P.log" 2>&1
  python3 - "$OUT/$P" "$P" <<'PY'
import csv, glob, os, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "bconv_fly" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {c: int(round(sum(v) / len(v))) for c, v in sorted(agg.items())})
PY
done
