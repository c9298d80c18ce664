#!/bin/bash
# Side build of ONE source of the library with extra flags -> tools/_build/lib_<tag>.so (full library, load through ND_LIB with
# tools/bench_ab.py / the *_bench.py tools).  Usage: tools/variant.sh <source stem> <tag> [extra hipcc flags...]
set -e
cd "$(dirname "$0")/.."
src=$1; tag=$2; shift 2
mkdir -p tools/_build
O=noisediff_amd/csrc/build
flags="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function"
case $src in
  conv3x3_wino2) flags="$flags -mllvm -pragma-unroll-threshold=1000000" ;;
  conv3x3_wino4) ;;
  *) flags="$flags -mllvm -amdgpu-mfma-vgpr-form=1" ;;
esac
hipcc $flags "$@" -c noisediff_amd/csrc/$src.hip -o tools/_build/${src}_$tag.o
objs=$(ls $O/*.o | grep -v "/$src.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/lib_$tag.so $objs tools/_build/${src}_$tag.o
