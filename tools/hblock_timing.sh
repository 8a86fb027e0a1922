#!/bin/bash
# Build a -DHB_TIMING variant of the library on the GPU box and print the per-phase stamps of the hierarchical-block kernel.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; V=/tmp/hbt; mkdir -p $V/obj
make -s -j8 -C "$R/binary-networks-pytorch_amd/csrc" OUTDIR=$V OBJDIR=$V/obj EXTRA=-DHB_TIMING $V/libbnn_hip.so 2>&1 | grep -E "error" 
BNN_AMD_LIB=$V/libbnn_hip.so python "$R/tools/exp_hblock_timing.py"
