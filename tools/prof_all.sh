set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01
rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra"
AOS2_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- $B > $O/stats1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -- $B > $O/stats3.log 2>&1
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
AOS2_CHUNKS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
AOS2_CHUNKS=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
AOS2_CHUNKS=1 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1
AOS2_CHUNKS=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
cd $R
for d in fetch write sq sq2; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); echo "== $d $f"; python tools/pmc_summary.py $f $O/pmc_$d.csv | head -8; done
for d in stats1 stats3; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$d.csv; head -12 $f; done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
du -sh $O
