cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" MF_SKIP_TRI MF_SKIP_MFMA PAIRS_SKIP_P2 PAIRS_SKIP_CHAIN; do
  if [ -n "$v" ]; then export VFR_LIB=$GRAFT_REPO_ROOT/video-fragments-retrieval_amd/lib/x_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_var_$v -- python tools/mfma_check.py 10000 5000 21 100 > gpurun_out/var_$v.txt 2>&1
  echo "== variant [$v]"; python tools/kstats.py $(ls gpurun_out/prof_var_$v/*/*kernel_stats.csv | head -1) 5 | grep -E "mfma_kernel<21, 8, 2, true, false>|pairs"
done
