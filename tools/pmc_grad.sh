#!/bin/bash
# PMC counters of the gradient kernels (separate passes, kernel-trace only):  gpurun -- 'bash tools/pmc_grad.sh'
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/pmc_grad"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
pmc() { n=$1; shift
  LIB=0 ONLY=${ONLY:-0,2,6} timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- python "$R/tools/bench_grad.py" > "$OUT/$n.log" 2>&1; }
pmc a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pmc b SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc c SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "grad_kernel" not in k: continue
        agg[(k.split("(")[0].replace("void bnn::", "")[:40], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-30s %14.0f  (n=%d)" % (c, sorted(v)[len(v)//2], len(v)))
PY
grep -il "error\|invalid" "$OUT"/*.log | head
