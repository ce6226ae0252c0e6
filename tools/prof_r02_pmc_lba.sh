# PMC passes for LocalBA: HBM bytes of the batch-32 kernels (separate FETCH / WRITE passes), f64 MFMA use of the single window
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02pmc
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/tools/gpu_lba_batch_prof.py > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/tools/gpu_lba_batch_prof.py > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU --output-format csv -d $O/mfma -- python $R/tools/gpu_lba_profile.py > $O/mfma.log 2>&1
cd $R
for d in fetch write mfma; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); echo "== $d"; python tools/pmc_summary2.py $f; done > $O/pmc_lba.txt 2>&1
cat $O/pmc_lba.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
