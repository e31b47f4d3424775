# SQ counters of the scorer kernels on the full-size check case (one pass; --kernel-trace only beside --pmc)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_cur
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_cur -- python tools/mfma_check.py 10000 5000 21 100 > gpurun_out/pmc_run.txt 2>&1
tail -3 gpurun_out/pmc_run.txt
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_cur/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'].split('(')[0][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
for k, c in agg.items():
    if 'score' in k or 'mfma' in k:
        print(k); print('   ', {n: f'{v:.4g}' for n, v in c.items()})
PY
