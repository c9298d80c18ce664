#!/bin/bash
# Diagnostic side builds of conv3x3_wino2.hip (W2_ABLATE bit mask) -> tools/_build/libw2_<mask>.so, full libraries that
# tools/wino_bench.py loads through ND_LIB.  Usage: tools/w2_variants.sh 0 1 2 4 8 ...
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
O=noisediff_amd/csrc/build
for m in "$@"; do
  hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -mllvm -pragma-unroll-threshold=1000000 -DW2_ABLATE=$m $W2_EXTRA -c noisediff_amd/csrc/conv3x3_wino2.hip -o tools/_build/w2_$m.o &
done
wait
for m in "$@"; do
  objs=$(ls $O/*.o | grep -v "conv3x3_wino2")
  hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libw2_$m.so $objs tools/_build/w2_$m.o
done
ls -la tools/_build/*.so
