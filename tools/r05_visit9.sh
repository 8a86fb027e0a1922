#!/bin/bash
# Round-5 GPU visit 9: the one-launch layer with a limit on how far waiting consumers pack ahead.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"; export TMPDIR=/tmp
V="$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants"
for rep in 1 2; do
for v in main w0 w2 w8; do
  env=(); [ $v != main ] && env=(BNN_AMD_LIB="$V/$v/libbnn_hip.so")
  echo "-- $v: $(env "${env[@]}" PLANS='1,56,16,2,-1;1,56,16,2,-1,-1,-1,1;1,56,16,2,-1,-1,-1,4' timeout 200 python tools/fly_quick.py 2>&1 | tail -1)"
done; done
echo "== fly tests w2"; BNN_AMD_LIB="$V/w2/libbnn_hip.so" timeout 900 python -m pytest tests/test_gpu_fly.py -q -x 2>&1 | tail -2
