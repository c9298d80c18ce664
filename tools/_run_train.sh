set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
mkdir -p gpurun_out/r3d
OUT=gpurun_out/r3d/r3j_training.txt
ERR=gpurun_out/r3d/r3j_training.stderr.log      # a failed benchmark leaves its reason here instead of an empty section
: > $ERR
echo "# tools/train_step_bench.py (PROFILE=0)" > $OUT
PROFILE=0 timeout -k 10 400 python3 tools/train_step_bench.py 2>> $ERR >> $OUT
echo "# tools/train_graph_bench.py" >> $OUT
timeout -k 10 300 python3 tools/train_graph_bench.py 2>> $ERR >> $OUT
echo "# NET=dropin tools/train_step_profile.py (noisediff_amd.NoiseDiffNet under autograd)" >> $OUT
NET=dropin timeout -k 10 200 python3 tools/train_step_profile.py 2>> $ERR >> $OUT
echo "# tools/train_step_profile.py, three runs" >> $OUT
for i in 1 2 3; do timeout -k 10 200 python3 tools/train_step_profile.py 2>> $ERR >> $OUT; done
echo "# ND_TRAIN_SPLITK=0 tools/train_step_profile.py (no split-K)" >> $OUT
ND_TRAIN_SPLITK=0 timeout -k 10 200 python3 tools/train_step_profile.py 2>> $ERR >> $OUT
echo "# ADAM_FUSED=1 tools/train_step_profile.py (torch.optim.Adam(fused=True))" >> $OUT
ADAM_FUSED=1 timeout -k 10 200 python3 tools/train_step_profile.py 2>> $ERR >> $OUT
rm -rf gpurun_out/r3d/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/prof -o train -- python3 tools/train_step_profile.py > gpurun_out/r3d/prof.log 2>&1
echo "# rocprofv3 --kernel-trace --stats -- python3 tools/train_step_profile.py (B=4 256x256, .hip()); tools/train_kernel_table.py <trace>" >> $OUT
python3 tools/train_kernel_table.py gpurun_out/r3d/prof/train_kernel_trace.csv >> $OUT
echo "# tools/train_bench.py (one layer, forward + backward, B=16 and B=4)" >> $OUT
timeout -k 10 300 python3 tools/train_bench.py 2>> $ERR >> $OUT
cat $OUT
