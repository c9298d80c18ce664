#!/bin/bash
# tools/wino4_bench.py against every tools/_build/libw4_*.so side build (same box, one after the other)
cd "$(dirname "$0")/.."
for lib in tools/_build/libw4_*.so; do
  echo "== $lib"
  ND_LIB=$lib timeout -k 10 120 python tools/wino4_bench.py 2>&1 | grep "wino4"
done
