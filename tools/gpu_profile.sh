#!/bin/bash
# Collect the artefacts quoted in DESIGN.md / bench.py: kernel-trace stats of the default bench command
# and PMC counters (separate passes) of the graded conv kernel.  Run on the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/gpu_profile.sh'
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/final"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 900 python "$R/bench.py" --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/bench.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/stats.log" 2>&1
pmc() { n=$1; shift
  ONLY=c2 ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- python "$R/tools/bench_conv.py" > "$OUT/$n.log" 2>&1; }
pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cut -c1-400 "$OUT/bench.json"; echo; head -8 "$OUT"/stats/*kernel_stats.csv | cut -c1-140
