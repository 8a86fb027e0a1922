#!/bin/bash
# one GPU visit: ablation variants of the stem kernel, interleaved twice
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
V=$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants
for rep in 1 2; do
  TAG=base timeout 120 python tools/stem_time.py 2>&1 | grep us
  for v in $(ls $V); do
    BNN_AMD_LIB=$V/$v/libbnn_hip.so TAG=$v timeout 120 python tools/stem_time.py 2>&1 | grep us
  done
done
