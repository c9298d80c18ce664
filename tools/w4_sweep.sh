#!/bin/bash
# tools/w4_clock.py over every tools/_build/lib_s_*.so (stamp builds of conv3x3_wino4 with one tuning define changed): one line per build and shape
cd "$(dirname "$0")/.."
for lib in tools/_build/lib_s_*.so; do
  echo "== $(basename $lib)"
  W4_SHAPES="${W4_SHAPES:-16,256,256,64,64;16,32,32,512,512}" ND_LIB=$lib timeout -k 10 120 python tools/w4_clock.py 2>&1 | grep -v amdgpu.ids | sed 's/ = stages.*all chunks/; all chunks/'
done
