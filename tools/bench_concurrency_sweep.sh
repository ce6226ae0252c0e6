# the composite with more LocalBA batches / steps in flight (AOS2_BENCH_LBA_HANDLES, AOS2_BENCH_INFLIGHT)
for cfg in "2 2" "3 2" "4 2" "3 3"; do set -- $cfg; echo "LBA_HANDLES=$1 INFLIGHT=$2"
  AOS2_BENCH_LBA_HANDLES=$1 AOS2_BENCH_INFLIGHT=$2 python bench.py --no-extra --no-cpu-baseline --no-verify --steps 40 > /tmp/sweep.json 2> /tmp/sweep.err || tail -3 /tmp/sweep.err
  python -c "import json; d=json.loads(open('/tmp/sweep.json').read().strip().splitlines()[-1]); print('   value %.0f ms_per_step %.3f' % (d['value'], d['ms_per_step']))"
done
