#!/bin/bash
# Variant builds of libbnn_hip.so that differ only in the compile flags of csrc/bconv_fly.hip (or of the sources named
# in VARSRCS, e.g. VARSRCS="bconv_fly.hip bconv.hip"); objects of the other sources are reused from build/obj:
#       bash tools/fly_variants.sh build name1="-DFOO=1" name2="-DBAR"
# Libraries land in binary-networks-pytorch_amd/bnn_amd/_lib/variants/<name>/libbnn_hip.so (git-ignored; they travel
# to the GPU box); select one with BNN_AMD_LIB.
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
C="$R/binary-networks-pytorch_amd/csrc"; O="$R/build/obj"; V="$R/binary-networks-pytorch_amd/bnn_amd/_lib/variants"
shift || true
make -s -C "$C" -j8 || exit 1
mkdir -p "$V"
for kv in "$@"; do
  k="${kv%%=*}"; flags="${kv#*=}"
  ( mkdir -p "$V/$k" "$R/build/obj_$k"; cp "$O"/*.o "$R/build/obj_$k/"; ok=1
    for src in ${VARSRCS:-bconv_fly.hip}; do
      /opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -c "$C/$src" -o "$R/build/obj_$k/${src%.hip}.o" || ok=0
    done
    [ $ok = 1 ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared "$R"/build/obj_$k/*.o -o "$V/$k/libbnn_hip.so" \
      && echo "built $k ($flags)" ) &
done
wait
rm -rf "$R"/build/obj_*
