#!/bin/bash
# Diagnostic side builds of conv3x3_wino4.hip (W4_ABLATE bit mask, W4_EXTRA flags) -> tools/_build/libw4_<mask>.so, full
# libraries that tools/wino4_bench.py loads through ND_LIB.  Usage: [W4_EXTRA=-D...] tools/w4_variants.sh 0 1 2 4 ...
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
O=noisediff_amd/csrc/build
for m in "$@"; do
  hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -DW4_ABLATE=$m $W4_EXTRA -c noisediff_amd/csrc/conv3x3_wino4.hip -o tools/_build/w4_$m.o &
done
wait
for m in "$@"; do
  objs=$(ls $O/*.o | grep -v "conv3x3_wino4")
  hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libw4_$m.so $objs tools/_build/w4_$m.o
done
