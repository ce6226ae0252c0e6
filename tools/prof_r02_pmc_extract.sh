# extractor alone at B = 512 (working set beyond the Infinity Cache): kernel stats + separate PMC passes -> gpurun_out/r02b512/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b512
rm -rf $O; mkdir -p $O
P="python $R/tools/prof_extract.py 512"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $P > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/FETCH_SIZE -- $P > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/WRITE_SIZE -- $P > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/SQ -- $P > $O/sq.log 2>&1
cd $R
for d in FETCH_SIZE WRITE_SIZE SQ; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python tools/pmc_summary2.py $f; done > $O/pmc_extract_b512.txt 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/extract_b512_kernel_stats.csv
python tools/pmc_extract_digest.py $O/pmc_extract_b512.txt $O/extract_b512_kernel_stats.csv > $O/extractor_counters.json
cat $O/pmc_extract_b512.txt; head -8 $O/extract_b512_kernel_stats.csv
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
