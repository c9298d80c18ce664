set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r1e_pytest.log 2>&1
python bench.py > gpurun_out/r1e_bench_default.jsonl 2> gpurun_out/r1e_bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1e -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-roofline > gpurun_out/prof_r1e.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-roofline --eager > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-roofline --eager > gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-roofline --eager > gpurun_out/pmc_mfma.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r1e_traffic.json "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-roofline --eager" > /dev/null
python tools/pmc_mfma.py gpurun_out/pmc_mfma gpurun_out/r1e_mfma_busy.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-roofline --eager"
cp $(ls gpurun_out/prof_r1e/*/*kernel_stats.csv | head -1) gpurun_out/r1e_rocprofv3_kernel_stats_bench_steps5.csv
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/prof_r1e
python bench.py --config cfg2 --no-cpu > gpurun_out/r1e_bench_cfg2.jsonl 2>/dev/null
python bench.py --config cfg4 --no-cpu > gpurun_out/r1e_bench_cfg4.jsonl 2>/dev/null
python tools/lsid_bench.py > gpurun_out/r1e_lsid.log 2>&1
tail -2 gpurun_out/r1e_pytest.log
