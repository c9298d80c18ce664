#!/bin/bash
# Where does conv3x3_wino2.hip need its `s_nop 1`?  Side builds with the wait states only in front of the MFMAs whose slice
# index is in W2_NOP_MASK (bit i = MFMA i of a step; what precedes MFMA i is slice i-1: 1..8 LDS reads, 3 the VALU clump +
# staging loads + LDS writes, 9..12 weight buffer loads, 13..15 and 0 nothing but the previous MFMA).
# (r2's probes also varied the pad of the zero-C MFMAs of a tile's first chunk separately: all identical, profiles/r2_wino2_nop_probe.txt.)
# Usage: tools/w2_nop_probe.sh 0x0000 0x1FFE ...   -> tools/_build/libw2nop_<mask>.so ; run tools/w2_nop_probe.py on the GPU box
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
O=noisediff_amd/csrc/build
for m in "$@"; do
  hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -mllvm -pragma-unroll-threshold=1000000 -DW2_NOP_MASK=$m $W2_EXTRA -c noisediff_amd/csrc/conv3x3_wino2.hip -o tools/_build/w2nop_$m.o &
done
wait
for m in "$@"; do
  objs=$(ls $O/*.o | grep -v "conv3x3_wino2")
  hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libw2nop_$m.so $objs tools/_build/w2nop_$m.o
done
ls -la tools/_build/libw2nop_*.so
