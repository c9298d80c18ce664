set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
OUT=gpurun_out/r3d/r3e_training_b.txt
rm -rf gpurun_out/r3d/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/prof -o train -- python3 tools/train_step_profile.py > gpurun_out/r3d/prof.log 2>&1
echo "# rocprofv3 --kernel-trace --stats -- python3 tools/train_step_profile.py (B=4 256x256, .hip()); tools/train_kernel_table.py" > $OUT
python3 tools/train_kernel_table.py gpurun_out/r3d/prof/train_kernel_stats.csv >> $OUT
echo "# tools/train_bench.py (one layer, forward + backward, B=16 and B=4)" >> $OUT
timeout -k 10 300 python3 tools/train_bench.py 2>/dev/null >> $OUT
cat $OUT
