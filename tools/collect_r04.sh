# copies the summaries of tools/prof_r04.sh from gpurun_out/prof_r04/ into profiles/
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof_r04
cp $O/bench_default.json profiles/r04_bench_n1.json
cp $O/bench_homogeneous.json profiles/r04_bench_n1_homogeneous_lba.json
cp $O/bench_kitti.json profiles/r04_bench_kitti.json
cp $O/bench_r03_form.json profiles/r04_bench_n1_r03_form.json
cp $O/bench_profiled.json profiles/r04_bench_profiled_run.json
cp $O/bench_kernel_stats.csv profiles/r04_bench_b512_kernel_stats.csv
cp $O/bench_fast_kernel_trace.txt profiles/r04_bench_fast_kernel_trace.txt
cp $O/kitti_kernel_stats.csv profiles/r04_bench_kitti_kernel_stats.csv
cp $O/lba_het64_kernel_stats.csv $O/lba_hom64_kernel_stats.csv profiles/ && mv profiles/lba_het64_kernel_stats.csv profiles/r04_lba_het64_kernel_stats.csv && mv profiles/lba_hom64_kernel_stats.csv profiles/r04_lba_hom64_kernel_stats.csv
cp $O/lba_batch32_kernel_stats.csv profiles/r04_lba_batch32_kernel_stats.csv
cp $O/lba_mix.txt profiles/r04_lba_mix.txt
cp $O/ldlt_phase_cycles.txt profiles/r04_ldlt_phase_cycles.txt
cp $O/desc_blur_ab.txt profiles/r04_desc_blur_ab.txt
cp $O/chain_latency.txt profiles/r04_chain_latency.txt
cp $O/pytest_gpu.log profiles/r04_pytest_gpu.log
[ -f $O/lba_unprofiled.txt ] && cp $O/lba_unprofiled.txt profiles/r04_lba_unprofiled.txt
