#!/bin/bash
# Build differently-configured libraries (in the container):   bash tools/conv_variants.sh build
# ... and time them on the GPU box:     gpurun -- 'bash tools/conv_variants.sh run'
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
V="$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants"
declare -A CFG=(
  [p2]="-DBNN_SGPR_PASSES=2"
  [p4]="-DBNN_SGPR_PASSES=4"
  [p8]="-DBNN_SGPR_PASSES=8"
)
if [ "${1:-build}" = "build" ]; then
  rm -rf "$V"; mkdir -p "$V"
  for k in "${!CFG[@]}"; do
    ( make -s -C "$R/binary-networks-pytorch_amd/csrc" OUTDIR="$V/$k" EXTRA="${CFG[$k]}" 2>&1 | grep -E "error" ) &
  done
  wait
  ls "$V"/*/
else
  for k in "${!CFG[@]}"; do
    echo "=== $k (${CFG[$k]})"
    [ -z "${STEM:-}" ] && BNN_AMD_LIB="$V/$k/libbnn_hip.so" ONLY="${ONLY:-}" timeout 300 python "$R/tools/bench_conv.py" 2>&1 | grep -v amdgpu.ids
    [ -n "${STEM:-}" ] && BNN_AMD_LIB="$V/$k/libbnn_hip.so" timeout 300 python "$R/tools/bench_stem.py" 2>&1 | grep "stem"
    [ -n "${NET:-}" ] && BNN_AMD_LIB="$V/$k/libbnn_hip.so" timeout 300 python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
  done
fi
