set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout -k 10 200 python3 tools/pack_compare.py tools/_build/lib_head_pack.so noisediff_amd/libnoisediff_hip.so > gpurun_out/r3d/pack_compare.log 2>&1 || { cat gpurun_out/r3d/pack_compare.log; exit 1; }
cat gpurun_out/r3d/pack_compare.log
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_hip_kernels.py -m gpu -x -q -k "dgrad or wino4 or conv3x3" > gpurun_out/r3d/pytest_train.log 2>&1 || { tail -30 gpurun_out/r3d/pytest_train.log; exit 1; }
tail -3 gpurun_out/r3d/pytest_train.log
timeout -k 10 200 python3 tools/train_step_profile.py > gpurun_out/r3d/train_step.log 2>&1; cat gpurun_out/r3d/train_step.log
rm -rf gpurun_out/r3d/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/prof -o train -- python3 tools/train_step_profile.py > gpurun_out/r3d/prof.log 2>&1
