#!/bin/bash
# PMC counters of the fused stem kernel (separate passes, kernel-trace only):  gpurun -- 'bash tools/pmc_stem.sh'
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/pmc_stem"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
pmc() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- env ONLY=default python "$R/tools/bench_stem.py" > "$OUT/$n.log" 2>&1; }
pmc a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pmc b SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc c SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
pmc d GRBM_GUI_ACTIVE
pmc e FETCH_SIZE WRITE_SIZE
pmc f TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum
pmc g TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
pmc h TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUFFER_WRITE_WAVEFRONTS_sum
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "stem_rows" not in k and "stem_split" not in k: continue
        agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-30s %14.0f  (n=%d)" % (c, sorted(v)[len(v)//2], len(v)))
PY
grep -il "error\|invalid" "$OUT"/*.log | head
