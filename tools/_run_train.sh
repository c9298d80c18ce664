set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_trainable.py tests/test_dropin.py -m gpu -x -q > gpurun_out/r3d/pytest_train.log 2>&1 || { tail -40 gpurun_out/r3d/pytest_train.log; exit 1; }
tail -2 gpurun_out/r3d/pytest_train.log
for i in 1 2 3; do timeout -k 10 200 python3 tools/train_step_profile.py 2>/dev/null | tail -1; done
rm -rf gpurun_out/r3d/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/prof -o train -- python3 tools/train_step_profile.py > gpurun_out/r3d/prof.log 2>&1
python3 tools/train_kernel_table.py gpurun_out/r3d/prof/train_kernel_trace.csv
