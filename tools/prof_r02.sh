set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
cd /tmp
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
tail -1 $O/stats.log
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/bench_kernel_stats.csv; head -40 $f
# the roofline launches inside the same trace: default bench.py (20 steps) so that bench_fast runs as in the bench line
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --no-cpu-baseline --no-extra > $O/trace.log 2>&1; cd $R
python tools/fast_kernel_from_trace.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/bench_fast_kernel_trace.txt; cat $O/bench_fast_kernel_trace.txt; tail -1 $O/trace.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py of this traced run: roofline.kernel_ms', d['roofline']['kernel_ms'])" >> $O/bench_fast_kernel_trace.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
du -sh $O
