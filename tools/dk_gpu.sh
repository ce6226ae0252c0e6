timeout 600 python -m pytest tests/test_extractor_gpu.py -m gpu -q -x 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/dk_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dk_stats -- python $GRAFT_REPO_ROOT/tools/prof_extract.py 512 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/dk_stats/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'aos2' in r['Name']: print(r['Name'][:40], r['Calls'], r['AverageNs'], r['MinNs'])
PY
