#!/bin/bash
# Collect the artefacts quoted in DESIGN.md / bench.py: bench lines, kernel-trace stats of the bench command (default:
# two batches in flight; and one batch in flight), PMC counters (separate passes, kernel-trace only) of the graded
# conv kernel and of the stem kernel.  Run on the GPU box:   gpurun --timeout 1500 -- 'bash tools/gpu_profile.sh'
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/final"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 900 python "$R/bench.py" --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/bench.json"
# full-size parity records (gpurun_out/c3_b256_parity_*.json) of the same build
( cd "$R" && timeout 600 python -m pytest tests/test_gpu_c3_full.py -q --timeout 300 2>&1 | tail -2 > "$OUT/c3_full.txt" )
timeout 300 python "$R/bench.py" --config c2 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/bench_c2.json"
timeout 600 python "$R/bench.py" --config c5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > "$OUT/bench_c5.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python "$R/bench.py" --steps 20 --warmup 5 --sustain 0 --no-extras --no-cpu-baseline > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1" -o bench1 -- python "$R/bench.py" --steps 20 --warmup 5 --spinup 200 --sustain 0 --streams 1 --no-extras --no-cpu-baseline --no-roofline > "$OUT/stats1.log" 2>&1   # (200 spin-up steps: the TRACE of this run is what tools/kernel_roofline.py reads, and traces above 8 MB are deleted below)
if [ "${QUICK:-0}" = "1" ]; then   # bench lines, kernel-trace stats and the c2 one-stream CSV only (no PMC, training, stem A/B)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2_1" -o bench_c2_1 -- python "$R/bench.py" --config c2 --steps 20 --warmup 5 --sustain 0 --no-extras --no-roofline --no-cpu-baseline > "$OUT/stats_c2_1.log" 2>&1
  find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
  cut -c1-300 "$OUT/bench.json"; echo; head -6 "$OUT"/stats_c2_1/*kernel_stats.csv | cut -c1-160; exit 0
fi
pmc() { n=$1; shift
  ONLY=c2 ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- python "$R/tools/bench_conv.py" > "$OUT/$n.log" 2>&1; }
pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
# the one-launch layer (bconv_fly_kernel) on config 2: the same counter passes, default plan
fpmc() { n=$1; shift
  ITERS=4 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/fly_$n" -o fly_$n -- python "$R/tools/run_fly.py" > "$OUT/fly_$n.log" 2>&1; }
fpmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
fpmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
fpmc grbm GRBM_GUI_ACTIVE
fpmc fetch FETCH_SIZE
fpmc write WRITE_SIZE
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2" -o bench_c2 -- python "$R/bench.py" --config c2 --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline > "$OUT/stats_c2.log" 2>&1
# the same line strictly one launch at a time and without the roofline block's packed-kernel / two-launch legs: the clean
# per-launch average of bconv_fly_kernel (the CSV above mixes in the two-stream leg)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2_1" -o bench_c2_1 -- python "$R/bench.py" --config c2 --steps 20 --warmup 5 --sustain 0 --no-extras --no-roofline --no-cpu-baseline > "$OUT/stats_c2_1.log" 2>&1
cp "$OUT/stats_c2_1.log" "$OUT/bench_c2_1stream.log" 2>/dev/null
spmc() { n=$1; shift
  ONLY=default timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/stem_$n" -o stem_$n -- python "$R/tools/bench_stem.py" > "$OUT/stem_$n.log" 2>&1; }
spmc a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
spmc b SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
spmc c SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
spmc d GRBM_GUI_ACTIVE
spmc e FETCH_SIZE
spmc f WRITE_SIZE
# training path: gradient kernels per layer (library backward beside them) and the whole step
{ timeout 250 python "$R/tools/bench_grad.py" 2>&1 | tail -8; timeout 250 env BATCH=256 python "$R/tools/bench_train.py" 2>&1 | tail -2;
  timeout 250 python "$R/tools/bench_train.py" 2>&1 | tail -1; } > "$OUT/train.txt" 2>&1
# counters of the gradient kernels (64->64 56x56 and 128->128 28x28) and the kernel breakdown of the last training step
( cd "$R" && ONLY=0,2 bash tools/pmc_grad.sh > "$OUT/grad_pmc.txt" 2>&1; bash tools/r04_train_prof.sh > /dev/null 2>&1; cp gpurun_out/r4train/last_step.txt "$OUT/train_last_step.txt" 2>/dev/null )
# roctx ranges of the C-ABI entry points in a marker trace (BNN_HIP_ROCTX=1)
BNN_HIP_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d "$OUT/roctx" -o roctx -- python "$R/bench.py" --engine fused --batch 32 --steps 3 --warmup 1 --spinup 2 --sustain 0 --no-extras --no-cpu-baseline --no-roofline > "$OUT/roctx.log" 2>&1
cd /tmp
# the two stem kernels side by side (bit-identity on ragged shapes, then timings)
timeout 300 python "$R/tools/stem_ab.py" 2>&1 | tail -12 > "$OUT/stem_ab.txt"
# VALU instructions per wave of every launch of one forward, from the SAME build (tools/kernel_roofline.py)
bash "$R/tools/pmc_net.sh" > "$OUT/pmc_net.log" 2>&1
# clocks / power under load and the two-stream timeline (round 5)
( cd "$R" && bash tools/power_and_overlap.sh > "$OUT/power.log" 2>&1 )
cd /tmp
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
cut -c1-400 "$OUT/bench.json"; echo; head -8 "$OUT"/stats1/*kernel_stats.csv 2>/dev/null | cut -c1-140
