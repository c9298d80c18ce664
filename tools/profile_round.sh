# Round profile set (run on the GPU box through gpurun): tests, bench lines, rocprofv3 kernel stats and the PMC passes the
# bench's roofline object cites.  usage: bash tools/profile_round.sh <tag> [a|b|all]     (e.g. r4b) -> gpurun_out/<tag>_*
# Part a: GPU tests, the default line, rocprofv3 --kernel-trace --stats of the same command.  Part b: the three PMC passes (FETCH_SIZE, WRITE_SIZE, matrix-pipe
# busy: separate runs, counters only), the other configurations (cfg2, cfg4, ref48 = the reference's shipped workload), --full, LSID, the 2-rank rehearsal.
# (a gpurun call is limited to 20 minutes: the two parts are two calls)
set -e
T=${1:-r5a}
PART=${2:-all}
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$R"
if [ "$PART" = a ] || [ "$PART" = all ]; then
[ -n "$SKIP_TESTS" ] || python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1 || { tail -30 gpurun_out/${T}_pytest.log; exit 1; }
python bench.py > gpurun_out/${T}_bench_default.jsonl 2> gpurun_out/${T}_bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$T -- python3 bench.py --steps 5 --warmup 1 --soak-s 0 --no-cpu --no-roofline --no-full-sample > gpurun_out/prof_$T.log 2>&1
cp $(ls gpurun_out/prof_$T/*/*kernel_stats.csv | head -1) gpurun_out/${T}_rocprofv3_kernel_stats_bench_steps5.csv
rm -rf gpurun_out/prof_$T
tail -2 gpurun_out/${T}_pytest.log
fi
if [ "$PART" = b ] || [ "$PART" = all ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --soak-s 0 --no-cpu --no-roofline --no-full-sample --eager > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --soak-s 0 --no-cpu --no-roofline --no-full-sample --eager > gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --soak-s 0 --no-cpu --no-roofline --no-full-sample --eager > gpurun_out/pmc_mfma.log 2>&1
CMD="rocprofv3 --kernel-trace --pmc {COUNTERS} --output-format csv -- python3 bench.py --steps 2 --warmup 1 --soak-s 0 --no-cpu --no-roofline --no-full-sample --eager"
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/${T}_traffic.json "${CMD/\{COUNTERS\}/FETCH_SIZE | WRITE_SIZE (separate passes)}" > /dev/null
python tools/pmc_mfma.py gpurun_out/pmc_mfma gpurun_out/${T}_mfma_busy.json "${CMD/\{COUNTERS\}/SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE}"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
python bench.py --config cfg2 --no-cpu > gpurun_out/${T}_bench_cfg2.jsonl 2>/dev/null
python bench.py --config cfg4 --no-cpu > gpurun_out/${T}_bench_cfg4.jsonl 2>/dev/null
python bench.py --config ref48 --steps 10 > gpurun_out/${T}_bench_ref48.jsonl 2>/dev/null
python bench.py --gpus 2 --one-device --backend gloo --no-cpu --steps 10 > gpurun_out/${T}_bench_2rank_selflaunch_one_device.jsonl 2>/dev/null
python bench.py --full --steps 1 --warmup 0 --no-cpu --no-roofline > gpurun_out/${T}_bench_full.jsonl 2>/dev/null
python tools/lsid_bench.py > gpurun_out/${T}_lsid.log 2>&1
for f in default cfg2 cfg4 ref48 full; do [ -f gpurun_out/${T}_bench_$f.jsonl ] && { echo "== $f"; python tools/bench_line.py < gpurun_out/${T}_bench_$f.jsonl; }; done
fi
