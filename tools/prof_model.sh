#!/bin/bash
# kernel-trace stats of one tools/bench_models.py family:  gpurun -- 'ONLY="resnet50" bash tools/prof_model.sh'
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/prof_model"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o m -- python "$R/tools/bench_models.py" > "$OUT/run.log" 2>&1
grep "img/s" "$OUT/run.log"
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-86s calls %5s avg %8.1f us  %5.1f%%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, 100.0 * int(r["TotalDurationNs"]) / tot))
PY
