#!/bin/bash
# Same-visit A/B of differently built libraries on the whole-net bench lines:
#   gpurun -- 'LIBS="base c1" bash tools/ab_libs.sh'      ("main" = the in-tree library; others under _lib/variants/<name>/)
# Two alternating rounds per library (clock ramp / box effects show up as round-to-round spread).
R="$(cd "$(dirname "$0")/.." && pwd)"
V="$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants"
one() {  # $1 = lib name, rest = bench args
  local lib="$1"; shift
  local env=(); [ "$lib" != main ] && env=(BNN_AMD_LIB="$V/$lib/libbnn_hip.so")
  env "${env[@]}" timeout 300 python "$R/bench.py" --no-cpu-baseline "$@" 2>/dev/null | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-6s %-22s %9.0f  1-batch %s  frac %s' % ('$lib', ' '.join(sys.argv[1:]) or 'resnet18', d['value'], d.get('one_batch_at_a_time', {}).get('value', d.get('two_batches_in_flight', {}).get('value')), d.get('roofline', {}).get('frac')))" "$@"
}
for rep in 1 2; do
  for lib in main ${LIBS:-base}; do
    one "$lib" --steps 30 --warmup 10 ${NET_ARGS:-}
    [ -n "${C2:-}" ] && [ $rep = 1 ] && one "$lib" --config c2 --steps 30 --warmup 10
    [ -n "${C5:-}" ] && one "$lib" --config c5 --steps 20 --warmup 5
  done
done
