#!/bin/bash
# Side build of the bf16 three-term F(4x4) experiment: tools/_build/lib_w4b_<tag>.so = the product library's objects + this kernel.
# usage: tools/experiments/wino4b/build.sh <tag> [extra hipcc flags...]
set -e
cd "$(dirname "$0")/../../.."
tag=$1; shift
mkdir -p tools/_build
hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Inoisediff_amd/csrc -Iinclude "$@" \
      -c tools/experiments/wino4b/conv3x3_wino4b.hip -o tools/_build/w4b_$tag.o
hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/lib_w4b_$tag.so noisediff_amd/csrc/build/*.o tools/_build/w4b_$tag.o
