#!/bin/bash
# Variant libraries of the stem kernel only (other objects are reused from build/obj):
#   tools/stem_variants.sh name1:"-DX=1" name2:"-DY" ...   ->  binary-networks-pytorch_amd/bnn_amd/_lib/variants/<name>/libbnn_hip.so
set -eu
cd "$(dirname "$0")/.."
make -C binary-networks-pytorch_amd/csrc -j8 > /dev/null
SRC=${SRC:-stem_rows}
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  mkdir -p build/obj_var binary-networks-pytorch_amd/bnn_amd/_lib/variants/$name
  /opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c binary-networks-pytorch_amd/csrc/$SRC.hip -o build/obj_var/${SRC}_$name.o
  objs=$(ls build/obj/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $objs build/obj_var/${SRC}_$name.o -o binary-networks-pytorch_amd/bnn_amd/_lib/variants/$name/libbnn_hip.so
  echo "built $name ($flags)"
done
