#!/bin/bash
# One GPU-box visit of round 2: smoke -> parity tests -> bench lines (c3, c2, c5) -> rocprofv3 kernel stats with ONE
# batch in flight.  Usage:  gpurun --timeout 1500 -- 'bash tools/r02_visit.sh <tag>'
set -u
TAG="${1:-a}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/r02$TAG"
mkdir -p "$OUT"
cd "$R"
export TMPDIR=/tmp
{ rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6; nproc; lscpu | grep -E "Model name|Socket" | head -3; } > "$OUT/box.txt" 2>&1
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -x -q -m gpu ${PYTEST_ARGS:-} 2>&1 | tail -25 | tee "$OUT/pytest_gpu.txt"
fi
echo "== bench c3"; timeout 600 python bench.py --steps 20 --warmup 5 2>"$OUT/bench_c3.err" | tail -1 | tee "$OUT/bench_c3.json" | cut -c1-1500
echo "== bench c2"; timeout 300 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline 2>"$OUT/bench_c2.err" | tail -1 | tee "$OUT/bench_c2.json" | cut -c1-600
echo "== bench c5"; timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>"$OUT/bench_c5.err" | tail -1 | tee "$OUT/bench_c5.json" | cut -c1-600
echo "== rocprof (one batch in flight)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o r02 -- python "$R/bench.py" --steps 20 --warmup 5 --streams 1 --no-extras --no-cpu-baseline --no-roofline > "$OUT/prof_run.log" 2>&1
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats_1stream.csv" && head -30 "$f" | cut -c1-200
# keep the merged-back payload small
find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
true
