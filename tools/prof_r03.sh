# Round-3 evidence in one gpurun call: gpu suite + smoke + default bench line, kernel stats of the bench, the roofline launches
# in a trace of the same command, extractor PMC passes at B = 512, LocalBA batch stats, single-frame chain, class-surface timing.
# Writes gpurun_out/prof_r03/ ; tools/collect_r03.sh copies the summaries into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
cd $R && timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
cd /tmp
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-verify"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/bench_kernel_stats.csv; head -30 $f | cut -c1-160
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --no-cpu-baseline --no-extra --no-verify > $O/trace.log 2>&1; cd $R
python tools/fast_kernel_from_trace.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/bench_fast_kernel_trace.txt; cat $O/bench_fast_kernel_trace.txt
tail -1 $O/trace.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py of this traced run: roofline.kernel_ms', d['roofline']['kernel_ms'])" >> $O/bench_fast_kernel_trace.txt
# extractor alone at B = 512: kernel stats + PMC passes (separate passes, no other trace domains)
cd /tmp
P="python $R/tools/prof_extract.py 512"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/xstats -- $P > $O/xstats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/FETCH_SIZE -- $P > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/WRITE_SIZE -- $P > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/SQ -- $P > $O/sq.log 2>&1
cd $R
for d in FETCH_SIZE WRITE_SIZE SQ; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python tools/pmc_summary2.py $f; done > $O/pmc_extract_b512.txt 2>&1
cp $(find $O/xstats -name "*kernel_stats.csv" | head -1) $O/extract_b512_kernel_stats.csv
python tools/pmc_extract_digest.py $O/pmc_extract_b512.txt $O/extract_b512_kernel_stats.csv | sed 's/r02_pmc_extract_b512/r03_pmc_extract_b512/; s/prof_r02_pmc_extract.sh/prof_r03.sh/' > $O/extractor_counters.json
cat $O/pmc_extract_b512.txt
# LocalBA batch (32 windows of 24 k edges), single-frame chain
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lbab -- python $R/tools/gpu_lba_batch_prof.py > $O/lba_batch.log 2>&1
cp $(find $O/lbab -name "*kernel_stats.csv" | head -1) $O/lba_batch_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kfw -- python $R/tools/gpu_keyframe_prof.py 2>&1 | grep "keyframe work" > $O/keyframe_work.txt
python $R/tools/kstats.py $O/kfw 8 >> $O/keyframe_work.txt
cd $R
python tools/gpu_chain_latency.py 2>&1 | grep -v "amdgpu.ids" > $O/chain_latency.txt
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain -- python $R/tools/gpu_chain_latency.py > /dev/null 2>&1; python $R/tools/kstats.py $O/chain 24 >> $O/chain_latency.txt; cd $R
rm -f $O/shim_timing.txt; AOS2_SHIM_TIMING_OUT=$O/shim_timing.txt python -m pytest tests/test_ref_signature_gpu.py -m gpu -q > /dev/null 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
du -sh $O; ls $O
