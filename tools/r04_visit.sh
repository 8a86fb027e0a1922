#!/bin/bash
# Round-4 GPU visit: new tests first (fail fast), the whole GPU suite, the bench line with all engines, linear shapes.
# Usage: gpurun --timeout 1500 -- 'bash tools/r04_visit.sh <tag>'
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG="${1:-r4}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== new tests"; timeout 1200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_dist.py -x -q 2>&1 | tail -30 | tee "$OUT/pytest_new.txt"
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_gpu_dropin.py --deselect tests/test_gpu_dist.py 2>&1 | tail -15 | tee "$OUT/pytest_gpu.txt"
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 > "$OUT/bench.json"; cut -c1-600 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench.json"))
    print("value", d["value"], "sustained", d.get("sustained"), "clock", d.get("engine_clock_mhz"))
    for k, v in d.get("engines", {}).items():
        print("  %-18s %9.0f img/s  %.3f ms  %s MHz" % (k, v["value"], v["ms_per_step"], v.get("engine_clock_mhz")))
    print("c1 gpu", d["cpu_baseline"]["c1_resnet18_32x32_b32"].get("gpu"))
    print("roofline", d["roofline"]["frac"], d["roofline"]["fp32_in_fp32_out"]["frac"])
except Exception as e:
    print("bench parse failed", e)
PY
echo "== linear shapes"; timeout 300 python tools/bench_linear.py 2>&1 | tail -14 | tee "$OUT/linear.txt"
