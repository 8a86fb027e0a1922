#!/bin/bash
# Round-5 GPU visit 7: stem with a deeper B-fragment prefetch (LEAN frees the registers): bit-identity, timings.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"; export TMPDIR=/tmp
V="$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants"
echo "== stem tests a2"; BNN_AMD_LIB="$V/a2/libbnn_hip.so" timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -q -k "stem" 2>&1 | tail -2
for v in main a2 main a2; do
  env=(); [ $v != main ] && env=(BNN_AMD_LIB="$V/$v/libbnn_hip.so")
  echo "-- stem alone $v"; env "${env[@]}" timeout 200 python tools/bench_stem.py 2>&1 | sed -n 2,2p
done
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'sustained', round(d.get('sustained',{}).get('value',0)))"; }
for v in main a2 main a2; do
  env=(); [ $v != main ] && env=(BNN_AMD_LIB="$V/$v/libbnn_hip.so")
  env "${env[@]}" timeout 300 python bench.py --steps 20 --warmup 5 --streams 1 --sustain 2 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | line "$v x1"
  env "${env[@]}" timeout 300 python bench.py --steps 20 --warmup 5 --sustain 2 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | line "$v x2"
done
