# correctness of every scoring mode, then a rocprofv3 kernel table of the full-size case
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/mfma_check.py > gpurun_out/mfma_check.txt 2>&1
grep -c "exact: True" gpurun_out/mfma_check.txt; tail -1 gpurun_out/mfma_check.txt
rm -rf gpurun_out/prof_cur
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cur -- python tools/mfma_check.py 10000 5000 21 100 > gpurun_out/mfma_prof.txt 2>&1
grep -A4 "^  exact" gpurun_out/mfma_prof.txt
python tools/kstats.py $(ls gpurun_out/prof_cur/*/*kernel_stats.csv | head -1) 9
