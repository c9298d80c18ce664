#!/bin/bash
# Same-box A/B of whole-step time: `tools/bench_ab.sh libA.so libB.so ...` runs bench.py (--no-cpu --no-roofline, 30 steps) against every library,
# ROUNDS times round-robin, and prints ms/step per run (boxes of the pool differ by ~5 %, so builds are only compared inside one call).
cd "$(dirname "$0")/.."
for r in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    ms=$(ND_LIB=$lib timeout -k 10 300 python tools/bench_ab.py --no-cpu --no-roofline --steps 30 --warmup 3 ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))")
    echo "round $r $(basename $lib) $ms ms/step"
  done
done
