# time the full-size check case with alternative libvfr builds (video-fragments-retrieval_amd/lib/x_<name>.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export VFR_LIB=$GRAFT_REPO_ROOT/video-fragments-retrieval_amd/lib/x_$v.so
  rm -rf gpurun_out/prof_var_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_var_$v -- python tools/mfma_check.py 10000 5000 21 100 > gpurun_out/var_$v.txt 2>&1
  echo "== variant [$v]"; grep "mfma ==" gpurun_out/var_$v.txt; python tools/kstats.py $(ls gpurun_out/prof_var_$v/*/*kernel_stats.csv | head -1) 6 | grep -E "mfma_kernel<21, 8, 2|pairs"
done
