#!/bin/bash
# First run on a node with more than one MI355X (none of the build's own boxes had two): in order, the one-GPU line, the two-device tests that are
# skipped on one GPU, then `bench.py --gpus 2 / 4 / 8`, and a comparison of every per-GPU rate with the one-GPU line.  One command, one log:
#   bash tools/first_node_check.sh [max_gpus] 2>&1 | tee gpurun_out/first_node_check.log
# Nothing here needs the reference; every rank uses its own GPU (RCCL over xGMI: one broadcast of the weights, no per-step collective).
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-$(python -c 'import torch; print(torch.cuda.device_count())')}
mkdir -p gpurun_out
echo "== devices visible: $N"
echo "== 1 GPU"
python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu | tee gpurun_out/first_node_n1.jsonl | python tools/bench_line.py || exit 1
if [ "$N" -ge 2 ]; then
  echo "== two physical devices in one process (nn.DataParallel device_ids=[0,1]) and the stream-device guards"
  python -m pytest tests/test_hip_net.py -q -m gpu -k "two_physical_devices or streams_device or device_ids" || exit 1
  echo "== two ranks, product path (weight broadcast over RCCL, sharded sampler, all-gather == one rank == oracle)"
  python -m pytest tests/test_hip_net.py -q -m gpu -k "n_rank or rccl" || exit 1
fi
for n in 2 4 8; do
  [ "$n" -le "$N" ] || break
  echo "== $n GPUs (weak scaling: 16 patches per GPU)"
  python bench.py --gpus $n --steps 20 --warmup 3 --no-cpu | tee gpurun_out/first_node_n$n.jsonl | python tools/bench_line.py || exit 1
done
python - <<'PY'
import glob, json
base = None
for f in sorted(glob.glob("gpurun_out/first_node_n[0-9]*.jsonl"), key=lambda s: int(s.split("_n")[-1].split(".")[0])):
    j = [json.loads(l) for l in open(f) if l.startswith("{")][-1]
    n = j["n_gpus"]
    base = base or j["value"] / n
    mg = j.get("multi_gpu") or {}
    print(f"n={n}: {j['value']:.4f} {j['unit']}  per GPU {j['value'] / n:.4f}  efficiency vs n=1 {j['value'] / n / base:.3f}  "
          f"broadcast {mg.get('broadcast_and_pack_ms')} ms / {mg.get('broadcast_bytes')} B  checksum equal {mg.get('arena_checksum_equal_on_all_ranks')}")
    # a slow RANK shows in its own GPU-event time, a slow BARRIER only in the wall clock of every rank
    if mg.get("ms_per_step_by_rank_gpu_events"):
        gpu, wall = mg["ms_per_step_by_rank_gpu_events"], mg["ms_per_step_by_rank_wall"]
        print("      ms/step by rank, GPU events: " + " ".join(f"{v:.2f}" for v in gpu) + f"   (spread {max(gpu) - min(gpu):.2f})")
        print("      ms/step by rank, wall clock: " + " ".join(f"{v:.2f}" for v in wall) + f"   (max wall - max GPU = {max(wall) - max(gpu):.2f}: barrier / launch side)")
PY
