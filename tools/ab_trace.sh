#!/bin/bash
# Per-launch durations of one forward (one batch in flight) for the in-tree library and variants, same visit:
#   gpurun -- 'LIBS="base u2" bash tools/ab_trace.sh'   then   python tools/ab_trace_cmp.py
R="$(cd "$(dirname "$0")/.." && pwd)"
V="$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants"
OUT="$R/gpurun_out/abtrace"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
run() { timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$1" -o t -- python "$R/bench.py" --steps 20 --warmup 5 --streams 1 --no-extras --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} > "$OUT/$1.log" 2>&1; }
run main
for l in ${LIBS:-base}; do BNN_AMD_LIB="$V/$l/libbnn_hip.so" run "$l"; done
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
ls "$OUT"
