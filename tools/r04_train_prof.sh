#!/bin/bash
# Kernel breakdown of the LAST training step (ResNet-18, batch 256) with the round-4 defaults.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/r4train"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
ONLY=mfma BATCH=256 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -o t -- python "$R/tools/bench_train.py" > "$OUT/log.txt" 2>&1
tail -1 "$OUT/log.txt"
f=$(find "$OUT/t" -name "*kernel_trace.csv" | head -1)
# window = the step time the run itself printed ("... gradient kernels 17.0 ms"), so that exactly one step is listed
MS=$(grep -oE "gradient kernels [0-9]+\.[0-9]+ ms" "$OUT/log.txt" | tail -1 | grep -oE "[0-9]+\.[0-9]+"); MS=${MS:-21.5}
python "$R/tools/last_step_profile.py" "$f" "$MS" multi_tensor 2>&1 | head -60 | tee "$OUT/last_step.txt"
find "$OUT" -name "*kernel_trace.csv" -size +6M -delete
