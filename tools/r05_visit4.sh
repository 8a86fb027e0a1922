#!/bin/bash
# Round-5 GPU visit 4: full GPU suite on the current build, bench A/B against the round-4 tree, one-stream traces.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/v4"; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee "$OUT/pytest_gpu.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('engines',{}); print('$1', round(d['value']), 'sustained', round(d.get('sustained',{}).get('value',0)), {k: round(v['value']) for k,v in e.items()}, 'frac', d.get('roofline',{}).get('frac'), 'fly', d.get('roofline',{}).get('fp32_in_fp32_out',{}).get('us'))"; }
echo "== bench main"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | tee "$OUT/bench_main.json" | line main
echo "== bench base r04"; ( cd build/base && timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | tee "$OUT/bench_base.json" | line base )
echo "== head"; timeout 120 python tools/bench_head.py 2>&1 | tail -2
cd /tmp
run() { timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tr_$1" -o t -- python "$2/bench.py" --steps 20 --warmup 5 --spinup 200 --sustain 0 --streams 1 --no-extras --no-cpu-baseline --no-roofline > "$OUT/tr_$1.log" 2>&1; }
run main "$R"; run base "$R/build/base"
for n in main base; do f=$(find "$OUT/tr_$n" -name "*kernel_stats.csv" | head -1); echo "-- $n"; python - "$f" <<'PY'
import csv,sys
tot=0
for r in list(csv.DictReader(open(sys.argv[1])))[:15]:
    name=r["Name"].replace("void bnn::","").split("(")[0][:72]
    print("%-74s %5s %9.1f us" % (name, r["Calls"], float(r["AverageNs"])/1e3))
PY
done
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
