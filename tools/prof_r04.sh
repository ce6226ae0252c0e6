# Round-4 evidence in one gpurun call.  Writes gpurun_out/prof_r04/ ; tools/collect_r04.sh copies the summaries into profiles/.
#   gpu suite + smoke; the default bench line (heterogeneous LocalBA windows, every frame pair distinct) and --lba-mix homogeneous;
#   ONE rocprofv3 run of bench.py that yields the kernel stats, the trace of the 20 solo FAST launches AND that run's own JSON line;
#   --workload kitti: line + kernel stats; LocalBA batches (het / hom 64 windows, r03's 32-window batch); LDLT phase cycles;
#   descriptor-stage A/B (per-keypoint blur vs whole-level blur); single-frame chain.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
cd $R && timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
python bench.py --lba-mix homogeneous --no-extra > $O/bench_homogeneous.json 2> $O/bench_homogeneous.err
# round 3's step exactly (32 distinct frame pairs tiled, 4 LocalBA problems of one size tiled, first-order Fuse targets only): like for like
AOS2_BENCH_SECOND_NEIGHBOURS=0 AOS2_BENCH_UNIQUE=32 python bench.py --lba-mix homogeneous --no-extra --no-cpu-baseline > $O/bench_r03_form.json 2> $O/bench_r03_form.err
python bench.py --workload kitti > $O/bench_kitti.json 2> $O/bench_kitti.err; tail -c 300 $O/bench_kitti.json
cd /tmp
# ---- one profiled run: stats + trace + the run's own line
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-extra > $O/bench_profiled.json 2> $O/bench_profiled.err
cd $R
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python tools/fast_kernel_from_trace.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/bench_fast_kernel_trace.txt
tail -1 $O/bench_profiled.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py line of this SAME profiled run: roofline.kernel_ms %.4f (frac %.4f), value %.0f frames/s, parity_checked.ok %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['value'], d['parity_checked']['ok']))" >> $O/bench_fast_kernel_trace.txt
cat $O/bench_fast_kernel_trace.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kitti -- python $R/bench.py --workload kitti --no-cpu-baseline --steps 20 > $O/kitti_profiled.json 2> $O/kitti_profiled.err
cp $(find $O/kitti -name "*kernel_stats.csv" | head -1) $O/kitti_kernel_stats.csv
# ---- LocalBA
cd $R
bash tools/prof_lba_mix.sh > $O/lba_mix.txt 2>&1
cp gpurun_out/prof_lba_mix/het_kernel_stats.csv $O/lba_het64_kernel_stats.csv; cp gpurun_out/prof_lba_mix/hom_kernel_stats.csv $O/lba_hom64_kernel_stats.csv
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lbab -- python $R/tools/gpu_lba_batch_prof.py > $O/lba_batch32.log 2>&1
cp $(find $O/lbab -name "*kernel_stats.csv" | head -1) $O/lba_batch32_kernel_stats.csv
cd $R
bash tools/build_ldlt_timing_lib.sh > /dev/null 2>&1
for n in 40 28 20; do N_LOCAL=$n AOS2_LIB=$R/active-orb-slam2_amd/lib/libaos2_ldlttiming.so AOS2_LBA_TRACE=1 python tools/gpu_ldlt_big.py 2>&1 | grep -E "reduced-system|diagonal blocks|n_local" | tail -3; done > $O/ldlt_phase_cycles.txt
# ---- descriptor stage A/B
(python tools/gpu_desc_blur_ab.py; AOS2_DESC_BLUR=level python tools/gpu_desc_blur_ab.py) 2>&1 | grep -v amdgpu.ids > $O/desc_blur_ab.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/blurA -- python $R/tools/gpu_desc_blur_ab.py > /dev/null 2>&1; python $R/tools/kstats.py $O/blurA 8 >> $O/desc_blur_ab.txt
AOS2_DESC_BLUR=level rocprofv3 --kernel-trace --stats --output-format csv -d $O/blurB -- python $R/tools/gpu_desc_blur_ab.py > /dev/null 2>&1; python $R/tools/kstats.py $O/blurB 8 >> $O/desc_blur_ab.txt
# ---- single-frame chain
cd $R
python tools/gpu_chain_latency.py 2>&1 | grep -v "amdgpu.ids" > $O/chain_latency.txt
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain -- python $R/tools/gpu_chain_latency.py > /dev/null 2>&1; python $R/tools/kstats.py $O/chain 24 >> $O/chain_latency.txt; cd $R
echo "# ---- ONE sequence with the next image's ExtractORB beside this frame's tracking (tools/gpu_pipelined_trace.py)" >> $O/chain_latency.txt
python tools/gpu_pipelined_trace.py 2>&1 | grep "per frame" >> $O/chain_latency.txt
# ---- LocalBA batches without the profiler
for g in "" 1; do for cfg in "het 64" "hom 64" "het 32" "hom 32"; do set -- $cfg; echo "AOS2_LBA_GROUPS=${g:-default} $(env ${g:+AOS2_LBA_GROUPS=$g} LBA_MIX=$1 LBA_N=$2 python tools/gpu_lba_mix_prof.py 2>&1 | grep windows | tail -1)"; done; done > $O/lba_unprofiled.txt
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
du -sh $O; ls $O
