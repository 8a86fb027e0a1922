#!/bin/bash
# Engine clock and socket power UNDER load (rocm-smi sampled every 250 ms beside the bench): one batch at a time, two
# batches in flight, the one-launch layer; then the two-stream timeline (tools/trace_overlap.py).  -> gpurun_out/power/
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/power"; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
sample() { # $1 = tag; samples while the command that follows runs
  tag=$1; shift
  ( while true; do rocm-smi -c -P --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > "$OUT/smi_$tag.jsonl" &
  SP=$!
  "$@" > "$OUT/run_$tag.log" 2>&1
  kill $SP 2>/dev/null; wait $SP 2>/dev/null
  python - "$OUT/smi_$tag.jsonl" "$tag" <<'PY'
import json, sys, re
rows = []
for ln in open(sys.argv[1]):
    try: d = json.loads(ln)
    except Exception: continue
    c = d.get("card0", {})
    rows.append(c)
if not rows: print(sys.argv[2], "no samples"); sys.exit()
keys = [k for k in rows[0] if "sclk" in k.lower() or "power" in k.lower() or "fclk" in k.lower()]
for k in keys:
    vals = []
    for r in rows:
        m = re.search(r"[-+]?\d*\.?\d+", str(r.get(k, "")).replace("(", " "))
        if m: vals.append(float(m.group()))
    if vals: print("%-10s %-45s n=%3d  min %8.1f  median %8.1f  max %8.1f" % (sys.argv[2], k, len(vals), min(vals), sorted(vals)[len(vals)//2], max(vals)))
PY
}
{
sample idle sleep 2
sample x1 python bench.py --steps 20 --warmup 5 --streams 1 --spinup 3000 --sustain 3 --no-cpu-baseline --no-extras --no-roofline
tail -1 "$OUT/run_x1.log" | cut -c1-200
sample x2 python bench.py --steps 20 --warmup 5 --spinup 3000 --sustain 3 --no-cpu-baseline --no-extras --no-roofline
tail -1 "$OUT/run_x2.log" | cut -c1-200
sample c2 python bench.py --config c2 --steps 20 --warmup 5 --spinup 8000 --sustain 3 --no-cpu-baseline --no-extras --no-roofline
tail -1 "$OUT/run_c2.log" | cut -c1-200
} 2>&1 | tee "$OUT/summary.txt"
echo "== two-stream timeline"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr2" -o t -- python "$R/bench.py" --steps 20 --warmup 5 --spinup 150 --sustain 0 --no-extras --no-cpu-baseline --no-roofline > "$OUT/tr2.log" 2>&1
f=$(find "$OUT/tr2" -name "*kernel_trace.csv" | head -1)
python "$R/tools/trace_overlap.py" "$f" 1400 | tee "$OUT/overlap.txt" | head -40
gzip -9 "$f"
