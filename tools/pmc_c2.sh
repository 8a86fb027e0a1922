#!/bin/bash
# PMC counters for the C2 conv kernel (separate passes; kernel-trace only, per gpurun's rules).
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/pmc"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  ONLY=c2 ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- python "$R/tools/bench_conv.py" > "$OUT/$n.log" 2>&1
  f=$(find "$OUT/$n" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'][:40]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'bconv' in k or 'pack_act' in k:
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'n=',len(next(iter(v.values()))))
PY
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
