#!/bin/bash
# Config 5 (ResNet(HBlock,[3,4,6,3]), batch 128): bench line, one-stream kernel trace (tools/kernel_roofline.py c5 reads
# it) and VALU counters of every launch of one forward.   gpurun --timeout 900 -- 'bash tools/c5_profile.sh'
# (BNN_AMD_FUSE_HBLOCK=0 C5OUT=c5_before: the launch-by-launch form of the hierarchical blocks, for the before / after table)
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/${C5OUT:-c5}"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 600 python "$R/bench.py" --config c5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > "$OUT/bench_c5.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1" -o c5 -- python "$R/bench.py" --config c5 --steps 20 --warmup 5 --spinup 100 --sustain 0 --streams 1 --no-extras --no-cpu-baseline --no-roofline > "$OUT/stats1.log" 2>&1
pmc() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$n" -o $n -- python "$R/bench.py" --config c5 --steps 3 --warmup 2 --spinup 20 --sustain 0 --streams 1 --engine fused --no-extras --no-cpu-baseline --no-roofline > "$OUT/pmc_$n.log" 2>&1; }
pmc a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
if [ "${C5_ONLY_TRAFFIC:-0}" = "1" ]; then rm -rf "$OUT/pmc_a"; fi
pmc b FETCH_SIZE
pmc c WRITE_SIZE
find "$OUT" -name "*kernel_trace.csv" -size +12M -delete
cut -c1-300 "$OUT/bench_c5.json"; echo; head -30 "$OUT"/stats1/*kernel_stats.csv 2>/dev/null | cut -c1-200
