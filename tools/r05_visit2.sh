#!/bin/bash
# Round-5 GPU visit 2: full GPU suite (all failures), stem register-diet variants beside the other stream's convolutions.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/v2"; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
V="$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants"
echo "== pytest -m gpu (main)"; timeout 1800 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_dist.py 2>&1 | tail -12 | tee "$OUT/pytest_gpu.txt"
echo "== pytest dist"; timeout 1500 python -m pytest tests/test_gpu_dist.py -q 2>&1 | tail -12 | tee "$OUT/pytest_dist.txt"
for v in lean lean1; do
  echo "== stem tests $v"; BNN_AMD_LIB="$V/$v/libbnn_hip.so" timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -q -k "stem" 2>&1 | tail -3
done
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s %s  %7.0f  sustained %7.0f' % ('$1', '$2', d['value'], d.get('sustained',{}).get('value',0)))"; }
for rep in 1 2; do
  for v in main k16 lean leanp lean1 leanp1; do
    env=(); [ $v != main ] && env=(BNN_AMD_LIB="$V/$v/libbnn_hip.so")
    env "${env[@]}" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | tee "$OUT/b_${v}_x2_$rep.json" | line $v x2
    [ $rep = 1 ] && env "${env[@]}" timeout 300 python bench.py --steps 20 --warmup 5 --streams 1 --sustain 0 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | tee "$OUT/b_${v}_x1.json" | line $v x1
  done
done
for v in main lean lean1; do
  env=(); [ $v != main ] && env=(BNN_AMD_LIB="$V/$v/libbnn_hip.so")
  echo "-- stem alone $v"; env "${env[@]}" timeout 200 python tools/bench_stem.py 2>&1 | sed -n 2,2p
done
