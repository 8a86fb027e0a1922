#!/bin/bash
# Kernel stats of the config-5 bench line, one batch in flight:  gpurun -- 'bash tools/prof_c5.sh'
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/prof_c5"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o c5 -- python "$R/bench.py" --config c5 --steps 20 --warmup 5 --streams 1 --no-extras --no-cpu-baseline --no-roofline > "$OUT/run.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    m = re.search(r"bnn::(\w+)(<[^>]*>)?", r["Name"])
    print("%-72s calls %5s avg %8.1f us %5.1f%%" % ((m.group(0) if m else r["Name"][:70]), r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
