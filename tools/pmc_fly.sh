#!/bin/bash
# PMC counters of the one-launch layer kernel (bconv_fly_kernel) on BASELINE config 2: separate passes, kernel-trace
# only (gpurun's rules).  Usage on the GPU box:  PLAN=1,56,16,2,-1 bash tools/pmc_fly.sh <tag>
R="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG="${1:-fly}"; OUT="$R/gpurun_out/pmc_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift
  ITERS=4 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- python "$R/tools/run_fly.py" > "$OUT/$n.log" 2>&1; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list); dur = []
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "bconv_fly" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(out, "sq1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "bconv_fly" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {c: int(round(sum(v) / len(v))) for c, v in sorted(agg.items())}
if dur:
    res["avg_duration_us_profiled"] = round(sum(dur) / len(dur), 1)
res["plan"] = os.environ.get("PLAN", "default")
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(res))
PY
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
