# copies the summaries of tools/prof_r03.sh from gpurun_out/prof_r03/ into profiles/
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof_r03
cp $O/bench_default.json profiles/r03_bench_n1.json
cp $O/bench_kernel_stats.csv profiles/r03_bench_b512_kernel_stats.csv
cp $O/bench_fast_kernel_trace.txt profiles/r03_bench_fast_kernel_trace.txt
cp $O/pmc_extract_b512.txt profiles/r03_pmc_extract_b512.txt
cp $O/extract_b512_kernel_stats.csv profiles/r03_extract_b512_kernel_stats.csv
cp $O/extractor_counters.json profiles/r03_extractor_counters.json
cp $O/lba_batch_kernel_stats.csv profiles/r03_lba_batch32_kernel_stats.csv
cp $O/chain_latency.txt profiles/r03_chain_latency.txt
cp $O/pytest_gpu.log profiles/r03_pytest_gpu.log
cp $O/keyframe_work.txt profiles/r03_keyframe_work.txt
mkdir -p profiles/r03_fuzz && cp gpurun_out/fuzz_r03/*.txt profiles/r03_fuzz/
(head -4 profiles/r03_shim_timing.txt | grep "^#"; cat $O/shim_timing.txt) > profiles/r03_shim_timing.txt.new && mv profiles/r03_shim_timing.txt.new profiles/r03_shim_timing.txt
