#!/bin/bash
# PMC counters (separate passes, kernel-trace only) of one bench_conv.py shape:
#   gpurun -- 'ONLY=l4_512 bash tools/pmc_shape.sh'      (BNN_AMD_LIB honoured)
R="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG="${ONLY:-c2}"; OUT="$R/gpurun_out/pmc_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
pmc() { n=$1; shift
  ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- python "$R/tools/bench_conv.py" > "$OUT/$n.log" 2>&1; }
pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAIT_INST_LDS
pmc sq3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH
pmc grbm GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bconv" not in k: continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-26s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
