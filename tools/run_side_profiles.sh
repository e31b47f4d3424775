# Side records of the round (one gpurun call): training step vs torch.nn, a serving request, the N-GPU rank step stand-in,
# the VGG extractor -- each with the tool's own output and, where useful, the rocprofv3 kernel table.
# usage: bash tools/run_side_profiles.sh <tag>   -> gpurun_out/<tag>_*.txt
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
{ echo "== python tools/train_bench.py 256 5"; python tools/train_bench.py 256 5 2>&1 | grep "training step"; echo "== python tools/train_bench.py 64 5"; python tools/train_bench.py 64 5 2>&1 | grep "training step"; } > $o/${tag}_train_bench.txt
rm -rf $o/prof_train; rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_train -- python tools/train_bench.py 256 5 > /dev/null 2>&1
{ echo "== rocprofv3 --kernel-trace --stats -- python tools/train_bench.py 256 5 (both arms in one process: vfr:: rows = the HIP autograd functions, Cijk_/MIOpen rows = torch.nn)"; python tools/kstats.py $(ls $o/prof_train/*/*kernel_stats.csv | head -1) 16; } >> $o/${tag}_train_bench.txt
{ echo "== python tools/small_batch.py 1 2 4 8 16 32 64 1024"; python tools/small_batch.py 1 2 4 8 16 32 64 1024 2>&1 | grep -v "amdgpu"; } > $o/${tag}_small_batch.txt
{ for n in 8 4 2; do echo "== python tools/rank_sim.py $n 20 1"; python tools/rank_sim.py $n 20 1 2>&1 | grep -v "amdgpu"; done; } > $o/${tag}_rank_sim.txt
{ echo "== python tools/vgg_bench.py"; python tools/vgg_bench.py 2>&1 | grep -v "amdgpu" | tail -8; } > $o/${tag}_vgg_bench.txt
tail -3 $o/${tag}_train_bench.txt; head -3 $o/${tag}_small_batch.txt; grep "N=" $o/${tag}_rank_sim.txt; head -3 $o/${tag}_vgg_bench.txt
