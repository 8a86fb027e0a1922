#!/bin/bash
# Round-5 GPU visit 6: head variants (LDS-staged means vs v_readlane broadcast), head tests on the variant, one-stream trace.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/v6"; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
V="$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants"
echo "== head main"; timeout 120 python tools/bench_head.py 2>&1 | tail -2
echo "== head rl"; BNN_AMD_LIB="$V/rl/libbnn_hip.so" timeout 120 python tools/bench_head.py 2>&1 | tail -2
echo "== head tests rl"; BNN_AMD_LIB="$V/rl/libbnn_hip.so" timeout 600 python -m pytest tests/test_gpu_fused.py -q -k "head" 2>&1 | tail -3
echo "== fused/dropin tests main (quick)"; timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_dropin.py tests/test_gpu_c3_full.py -q -x 2>&1 | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'sustained', round(d.get('sustained',{}).get('value',0)))"; }
for v in main rl; do
  env=(); [ $v != main ] && env=(BNN_AMD_LIB="$V/$v/libbnn_hip.so")
  env "${env[@]}" timeout 300 python bench.py --steps 20 --warmup 5 --streams 1 --sustain 2 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | line "$v x1"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tr_main" -o t -- python "$R/bench.py" --steps 20 --warmup 5 --spinup 200 --sustain 0 --streams 1 --no-extras --no-cpu-baseline --no-roofline > "$OUT/tr_main.log" 2>&1
f=$(find "$OUT/tr_main" -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:17]:
    name=r["Name"].replace("void bnn::","").split("(")[0][:72]
    print("%-74s %5s %9.1f us" % (name, r["Calls"], float(r["AverageNs"])/1e3))
PY
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
