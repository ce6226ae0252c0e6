# counters of describe_kernel at B = 512 (debug aid for the descriptor kernel): two passes
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; P="python $R/tools/prof_extract.py 512"
rm -rf $R/gpurun_out/dk_sq $R/gpurun_out/dk_sq2 $R/gpurun_out/dk_sq3
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/dk_sq -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/dk_sq2 -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/dk_sq3 -- $P > /dev/null 2>&1
cd $R; for d in dk_sq dk_sq2 dk_sq3; do python tools/pmc_summary2.py $(find gpurun_out/$d -name "*counter_collection.csv" | head -1) | grep "describe"; done
find gpurun_out/dk_sq gpurun_out/dk_sq2 gpurun_out/dk_sq3 -name "*counter_collection.csv" -delete
