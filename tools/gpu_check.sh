#!/bin/bash
# One GPU-box visit: smoke -> parity tests -> bench -> rocprofv3 kernel stats.
# Usage (from the build container):  gpurun --timeout 1500 -- 'bash tools/gpu_check.sh'
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out"
mkdir -p "$OUT"
cd "$R"
export TMPDIR=/tmp
{
  echo "== rocminfo (clocks/CUs)"; rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|Wavefront Size|gfx" | head -12
  echo "== host"; nproc; lscpu | grep -E "Model name|Socket|Core|Thread" | head -6
} > "$OUT/box.txt" 2>&1

echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee "$OUT/pytest_gpu.txt"
echo "== bench (graph)"; timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee "$OUT/bench.json"
echo "== bench (fused eager)"; timeout 600 python bench.py --steps 10 --warmup 3 --engine fused --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-260
echo "== bench (layerwise)"; timeout 600 python bench.py --steps 10 --warmup 3 --engine layerwise --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-260
echo "== rocprof"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o r01 -- python "$R/bench.py" --steps 5 --warmup 2 --engine fused --no-cpu-baseline --no-roofline > "$OUT/prof_run.log" 2>&1
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -16 "$f" | cut -c1-150
