#!/bin/bash
# PMC counters of every kernel of one ResNet-18 forward (one batch in flight, kernel-trace only):
#   gpurun -- 'bash tools/pmc_net.sh'   ->  gpurun_out/pmc_net/  (summarised by tools/collect_net_pmc.py)
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/pmc_net"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
pmc() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- python "$R/bench.py" --steps 3 --warmup 2 --spinup 20 --sustain 0 --streams 1 --engine fused --no-extras --no-cpu-baseline --no-roofline > "$OUT/$n.log" 2>&1; }
pmc a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pmc b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
ls "$OUT"/*/ | head; grep -il "error" "$OUT"/*.log | head
