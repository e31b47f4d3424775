# The round's profile set, one gpurun call: bench (unprofiled, with CPU baseline + sub-records), the same command under
# rocprofv3 --kernel-trace --stats, the n = 6 bench, and the PMC passes (FETCH_SIZE / WRITE_SIZE separately, MFMA busy).
# usage: bash tools/run_round_profiles.sh <tag>
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag; rm -rf $o; mkdir -p $o
python bench.py --steps 20 --warmup 3 > $o/bench_unprofiled.json 2> $o/bench_unprofiled.err && echo "bench ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $o/bench.json 2> $o/bench.err && echo "stats ok"
cp $(ls $o/stats/*/*kernel_stats.csv | head -1) $o/kernel_stats.csv
python bench.py --steps 5 --warmup 2 --clips 6 --no-cpu-baseline --no-extras > $o/bench_n6.json 2> $o/bench_n6.err && echo "n6 ok"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $o/pmc_fetch.json 2> $o/pmc_fetch.err && echo "fetch ok"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $o/pmc_write.json 2> $o/pmc_write.err && echo "write ok"
python tools/pmc_traffic.py $o/pmc_fetch $o/pmc_write $o/pmc_traffic.json "profiles/${tag}_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) of bench.py --steps 2 --warmup 1, bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; a committed profile, not measured in this run" > /dev/null && echo "traffic ok"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $o/pmc_sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $o/pmc_sq.json 2> $o/pmc_sq.err && echo "sq ok"
python tools/pmc_sq_summary.py "$o"
