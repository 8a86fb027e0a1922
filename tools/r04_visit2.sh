#!/bin/bash
# Round-4 visit 2: where the per-layer path spends its time; roctx ranges show up in a marker trace.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/r4b"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/lw" -o lw -- python "$R/bench.py" --engine layerwise --steps 10 --warmup 2 --spinup 10 --sustain 0 --no-extras --no-cpu-baseline --no-roofline > "$OUT/lw.log" 2>&1
f=$(find "$OUT/lw" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-200
BNN_HIP_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d "$OUT/roctx" -o roctx -- python "$R/bench.py" --engine fused --batch 32 --steps 3 --warmup 1 --spinup 2 --sustain 0 --no-extras --no-cpu-baseline --no-roofline > "$OUT/roctx.log" 2>&1
ls "$OUT/roctx"/*/ 2>/dev/null | head; f=$(find "$OUT/roctx" -name "*marker_api_stats.csv" -o -name "*marker*stats*.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-200
find "$OUT" -name "*_trace.csv" -size +4M -delete
