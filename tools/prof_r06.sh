# Round-6 evidence in one gpurun call (PARTS selects: default all).  Writes gpurun_out/prof_r06/ ; tools/collect_r06.sh copies the summaries into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r06
mkdir -p $O
PARTS=${PARTS:-"suite bench prof kitti pmc_extract pmc_kitti pmc_lba lba chain"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
cd $R
if has suite; then
  timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
fi
if has bench; then
  # the default line ten times in a row (the spread), the median run is the committed line; then the other forms once each
  for i in 01 02 03 04 05; do python bench.py > $O/bench_default_$i.json 2> $O/bench_default_$i.err; tail -c 200 $O/bench_default_$i.json; done
  python bench.py --host-images --no-extra > $O/bench_host_images.json 2> $O/bench_host_images.err
  python bench.py --lba-mix homogeneous --no-extra > $O/bench_homogeneous.json 2> $O/bench_homogeneous.err
  AOS2_BENCH_RUNNER=python AOS2_BENCH_LBA_HANDLES=2 python bench.py --no-extra --no-cpu-baseline > $O/bench_python_threads.json 2> $O/bench_python_threads.err
fi
cd /tmp
if has prof; then
  # ---- one profiled run: stats + trace + the run's own line
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-extra > $O/bench_profiled.json 2> $O/bench_profiled.err
  cd $R
  cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
  python tools/fast_kernel_from_trace.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/bench_fast_kernel_trace.txt
  tail -1 $O/bench_profiled.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py line of this SAME profiled run: roofline.kernel_ms %.4f (frac %.4f), value %.0f frames/s, parity_checked.ok %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['value'], d['parity_checked']['ok']))" >> $O/bench_fast_kernel_trace.txt
  cat $O/bench_fast_kernel_trace.txt; head -14 $O/bench_kernel_stats.csv | cut -c1-150
  cd /tmp
fi
if has kitti; then
  python $R/bench.py --workload kitti > $O/bench_kitti.json 2> $O/bench_kitti.err; tail -c 200 $O/bench_kitti.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kitti -- python $R/bench.py --workload kitti --no-cpu-baseline --steps 20 > $O/kitti_profiled.json 2> $O/kitti_profiled.err
  cp $(find $O/kitti -name "*kernel_stats.csv" | head -1) $O/kitti_kernel_stats.csv
fi
if has pmc_extract; then
  # ---- counters of the extractor at B = 512 (beyond the Infinity Cache): separate passes
  P="python $R/tools/prof_extract.py 512"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_stats -- $P > $O/x_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/x_FETCH -- $P > $O/x_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/x_WRITE -- $P > $O/x_write.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/x_SQ -- $P > $O/x_sq.log 2>&1
  cd $R
  for d in x_FETCH x_WRITE x_SQ; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python tools/pmc_summary2.py $f; done > $O/pmc_extract_b512.txt 2>&1
  cp $(find $O/x_stats -name "*kernel_stats.csv" | head -1) $O/extract_b512_kernel_stats.csv
  python tools/pmc_extract_digest.py $O/pmc_extract_b512.txt $O/extract_b512_kernel_stats.csv r06 > $O/extractor_counters.json
  head -40 $O/pmc_extract_b512.txt
  cd /tmp
fi
if has pmc_kitti; then
  # ---- the same passes for the KITTI extraction (1241 x 376, 2000 features, 256 images = 128 stereo frames' eyes... one eye per image)
  P="python $R/tools/prof_extract.py 256 kitti"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_stats -- $P > $O/k_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/k_FETCH -- $P > $O/k_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/k_WRITE -- $P > $O/k_write.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/k_SQ -- $P > $O/k_sq.log 2>&1
  cd $R
  for d in k_FETCH k_WRITE k_SQ; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python tools/pmc_summary2.py $f; done > $O/pmc_extract_kitti_b256.txt 2>&1
  cp $(find $O/k_stats -name "*kernel_stats.csv" | head -1) $O/extract_kitti_b256_kernel_stats.csv
  python tools/pmc_extract_digest.py $O/pmc_extract_kitti_b256.txt $O/extract_kitti_b256_kernel_stats.csv r06 256 > $O/extractor_counters_kitti.json
  head -12 $O/pmc_extract_kitti_b256.txt
  cd /tmp
fi
if has pmc_lba; then
  # ---- counters of the mixed 64-window LocalBA batch: HBM bytes per kernel, f64 MFMA use of the reduced-system kernel
  L="env LBA_MIX=het LBA_N=64 python $R/tools/gpu_lba_mix_prof.py"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/l_stats -- $L > $O/l_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/l_FETCH -- $L > $O/l_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/l_WRITE -- $L > $O/l_write.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU --output-format csv -d $O/l_MFMA -- $L > $O/l_mfma.log 2>&1
  cd $R
  for d in l_FETCH l_WRITE l_MFMA; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); echo "== $d"; python tools/pmc_summary2.py $f; done > $O/pmc_lba.txt 2>&1
  cp $(find $O/l_stats -name "*kernel_stats.csv" | head -1) $O/lba_het64_kernel_stats.csv
  python tools/pmc_lba_digest.py $O/pmc_lba.txt $O/lba_het64_kernel_stats.csv > $O/lba_counters.json
  head -60 $O/pmc_lba.txt; head -12 $O/lba_het64_kernel_stats.csv | cut -c1-150; head -40 $O/lba_counters.json
  cd /tmp
fi
cd $R
if has lba; then
  for o in "" old; do for cfg in "het 64" "hom 64" "het 32" "hom 32"; do set -- $cfg; echo "AOS2_LDLT=${o:-new} $(env ${o:+AOS2_LDLT=$o} LBA_MIX=$1 LBA_N=$2 python tools/gpu_lba_mix_prof.py 2>&1 | grep windows | tail -1)"; done; done > $O/lba_unprofiled.txt
  cat $O/lba_unprofiled.txt
  python tools/gpu_lba_determinism.py 1200 2>&1 | tail -3 > $O/lba_determinism.txt; cat $O/lba_determinism.txt
fi
if has ldlt; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I active-orb-slam2_amd/csrc tools/microbench/ldlt_reg_bench.hip -o /tmp/ldlt_reg_bench && /tmp/ldlt_reg_bench 38 > $O/ldlt_phase_cycles.txt 2>&1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/microbench/lr_latency.hip -o /tmp/lr_latency && /tmp/lr_latency > $O/lr_latency.txt 2>&1
  cat $O/ldlt_phase_cycles.txt | cut -c1-260; cat $O/lr_latency.txt
fi
if has chain; then
  python tools/gpu_chain_latency.py 2>&1 | grep -v "amdgpu.ids" > $O/chain_latency.txt
  python tools/gpu_pipelined_trace.py 2>&1 | grep "per frame" >> $O/chain_latency.txt
  python tools/gpu_graph_chain.py 2>&1 | grep "per frame\|nodes\|matches" >> $O/chain_latency.txt
  tail -5 $O/chain_latency.txt
fi
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete; find $O -name "*counter_collection.csv" -delete
du -sh $O; ls $O
