R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/c5x2"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats2" -o c5 -- python "$R/bench.py" --config c5 --steps 20 --warmup 5 --spinup 100 --sustain 0 --no-extras --no-cpu-baseline --no-roofline > "$OUT/stats2.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -size +12M -delete
tail -1 "$OUT/stats2.log" | cut -c1-200
