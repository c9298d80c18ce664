#!/bin/bash
# Same-box A/B of whole-step time over values of ONE environment knob: `tools/bench_env_ab.sh ND_W4_STREAM_MB 0 48 100 100000`
cd "$(dirname "$0")/.."
knob=$1; shift
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    ms=$(env $knob=$v timeout -k 10 300 python bench.py --no-cpu --no-roofline --steps 30 --warmup 3 ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))")
    echo "round $r $knob=$v $ms ms/step"
  done
done
