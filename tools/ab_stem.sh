#!/bin/bash
# Same-visit A/B of the stem kernel alone:  gpurun -- 'LIBS="p1 p2" bash tools/ab_stem.sh'   (main = in-tree library)
R="$(cd "$(dirname "$0")/.." && pwd)"
V="$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants"
for rep in 1 2; do
  for lib in main ${LIBS:-}; do
    env=(); [ "$lib" != main ] && env=(BNN_AMD_LIB="$V/$lib/libbnn_hip.so")
    echo "$lib: $(env "${env[@]}" ONLYMODE=default python "$R/tools/bench_stem.py" 2>/dev/null | head -1)"
  done
done
