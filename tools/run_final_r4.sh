# The final record of round 4 in one gpurun call: GPU test suite, the round's profile set (run_round_profiles.sh), the side
# records (run_side_profiles.sh), the scorer A/B with the round's switches all off / all on, the RCCL world-size-1 rehearsal,
# ResNet-152, the graph-replay comparison.  usage: bash tools/run_final_r4.sh <tag>
tag=${1:-r4z}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $o/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -2 $o/${tag}_tests.log
bash tools/run_round_profiles.sh $tag > $o/${tag}_round.log 2>&1; grep -E "ok$" $o/${tag}_round.log | tr '\n' ' '; echo
VFR_ONE_GPU_MS=$(python -c "import json;print(round(json.load(open('$o/$tag/bench_unprofiled.json'))['ms_per_step'],2))") bash tools/run_side_profiles.sh $tag > $o/${tag}_side.log 2>&1; grep "N=" $o/${tag}_rank_sim.txt
timeout -k 10 300 python tools/scorer_ab.py --zip score_defer=-1,8 score_sort=0,1 score_hist=0,1 score_pre_b=625,0 > $o/${tag}_scorer_ab.txt 2>&1; grep -E "bench|planted" $o/${tag}_scorer_ab.txt | cut -c1-120
timeout -k 10 300 python tools/scorer_ab.py --clips 6 --zip score_defer=-1,8 score_sort=0,1 score_hist=0,1 score_pre_b=625,0 > $o/${tag}_scorer_ab_n6.txt 2>&1
VFR_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $o/${tag}_bench_rccl_world1.json 2> $o/${tag}_bench_rccl_world1.err; python tools/show_line.py $o/${tag}_bench_rccl_world1.json
timeout -k 10 200 python tools/resnet_bench.py 150 3 > $o/${tag}_resnet_bench.txt 2>&1; head -1 $o/${tag}_resnet_bench.txt
timeout -k 10 200 python tools/graph_request.py 1 8 32 64 > $o/${tag}_graph_request.txt 2>&1; cat $o/${tag}_graph_request.txt | grep Nq
