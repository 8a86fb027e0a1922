#!/bin/bash
# Round-5 GPU visit 1: full GPU suite on the new build, stem variant check, whole-net A/B against the round-4 tree
# (build/base: git worktree of the round-4 head with its own library), head kernels, per-launch traces.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/v1"; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
V="$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants"
echo "== pytest -m gpu (main)"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee "$OUT/pytest_gpu.txt"
if [ -f "$V/k16/libbnn_hip.so" ]; then
  echo "== stem tests with the 16x16x16 tail (bit-identity against the round-2 kernel)"
  BNN_AMD_LIB="$V/k16/libbnn_hip.so" timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -q -k "stem" 2>&1 | tail -8 | tee "$OUT/pytest_k16.txt"
fi
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('engines',{}); print('$1', round(d['value']), 'sustained', round(d.get('sustained',{}).get('value',0)), {k: round(v['value']) for k,v in e.items()}, 'frac', d.get('roofline',{}).get('frac'), 'fly', d.get('roofline',{}).get('fp32_in_fp32_out',{}).get('us'))"; }
for rep in 1 2; do
  echo "== bench main (rep $rep)"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $([ $rep = 2 ] && echo --no-extras --no-roofline) 2>/dev/null | tail -1 | tee "$OUT/bench_main_$rep.json" | line main
  echo "== bench base r04 (rep $rep)"; ( cd build/base && timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $([ $rep = 2 ] && echo --no-extras --no-roofline) 2>/dev/null | tail -1 | tee "$OUT/bench_base_$rep.json" | line base )
  if [ -f "$V/k16/libbnn_hip.so" ]; then
    echo "== bench k16 (rep $rep)"; BNN_AMD_LIB="$V/k16/libbnn_hip.so" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | tee "$OUT/bench_k16_$rep.json" | line k16
  fi
done
echo "== head"; timeout 120 python tools/bench_head.py 2>&1 | tail -6 | tee "$OUT/head.txt"
echo "== stem"; timeout 200 python tools/bench_stem.py 2>&1 | tail -8 | tee "$OUT/stem.txt"
[ -f "$V/k16/libbnn_hip.so" ] && BNN_AMD_LIB="$V/k16/libbnn_hip.so" timeout 200 python tools/bench_stem.py 2>&1 | tail -8 | tee "$OUT/stem_k16.txt"
echo "== traces"
cd /tmp
run() { # $1 = name, $2 = repo root
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tr_$1" -o t -- python "$2/bench.py" --steps 20 --warmup 5 --spinup 200 --sustain 0 --streams 1 --no-extras --no-cpu-baseline --no-roofline > "$OUT/tr_$1.log" 2>&1; }
run main "$R"; run base "$R/build/base"
for n in main base; do f=$(ls "$OUT"/tr_$n/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(find "$OUT/tr_$n" -name "*kernel_stats.csv" | head -1); echo "-- $n"; head -24 "$f" | cut -c1-170; done
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
