# copies the summaries of tools/prof_r02.sh, prof_r02_extra.sh and prof_r02_pmc_extract.sh from gpurun_out/ into profiles/
set -e
cd "$(dirname "$0")/.."
O=gpurun_out
cp $O/prof_r02/bench_default.json profiles/r02_bench_n1.json
cp $O/prof_r02/bench_kernel_stats.csv profiles/r02_bench_b256_kernel_stats.csv
cp $O/prof_r02/bench_fast_kernel_trace.txt profiles/r02_bench_fast_kernel_trace.txt
cp $O/prof_r02x/lba_single_kernel_stats.csv profiles/r02_lba_kernel_stats.csv
cp $O/prof_r02x/lba_batch_kernel_stats.csv profiles/r02_lba_batch32_kernel_stats.csv
for f in lba_host_phases chain_latency po_phase_cycles ldlt_phase_cycles shim_timing host_api_latency; do cp $O/prof_r02x/$f.txt profiles/r02_$f.txt; done
cp $O/r02b512/pmc_extract_b512.txt profiles/r02_pmc_extract_b512.txt
cp $O/r02b512/extract_b512_kernel_stats.csv profiles/r02_extract_b512_kernel_stats.csv
cp $O/r02b512/extractor_counters.json profiles/r02_extractor_counters.json
