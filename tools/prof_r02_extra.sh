# round-2 evidence beyond the bench line: LocalBA kernel stats / host phases, single-frame chain, phase cycle counters,
# shim timings.  Run on the GPU box through gpurun; writes gpurun_out/prof_r02x/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02x
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lba -- python $R/tools/gpu_lba_profile.py > $O/lba_single.log 2>&1
python $R/tools/kstats.py $O/lba 16 > $O/lba_single_kernel_stats.txt; cp $(find $O/lba -name "*kernel_stats.csv" | head -1) $O/lba_single_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lbab -- python $R/tools/gpu_lba_batch_prof.py > $O/lba_batch.log 2>&1
python $R/tools/kstats.py $O/lbab 16 > $O/lba_batch_kernel_stats.txt; cp $(find $O/lbab -name "*kernel_stats.csv" | head -1) $O/lba_batch_kernel_stats.csv
cd $R
AOS2_LBA_PROF=1 python tools/gpu_lba_profile.py --batch > $O/lba_host_phases.txt 2>&1
python tools/gpu_lba_windows.py >> $O/lba_host_phases.txt 2>&1
python tools/gpu_chain_latency.py 2>&1 | grep -v "^Traceback\|File\|Attribute\|Exception" > $O/chain_latency.txt
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain -- python $R/tools/gpu_chain_latency.py > /dev/null 2>&1; python $R/tools/kstats.py $O/chain 30 >> $O/chain_latency.txt; cd $R
AOS2_LIB=$R/active-orb-slam2_amd/lib/libaos2_potiming.so python tools/dbg/po_time.py 2>&1 | grep "^PO" | sort -u > $O/po_phase_cycles.txt
AOS2_LBA_TRACE=1 AOS2_LIB=$R/active-orb-slam2_amd/lib/libaos2_ldlttiming.so python tools/dbg/lba_trace.py 2>&1 | grep "reduced-system\|diagonal" | tail -2 > $O/ldlt_phase_cycles.txt
python -m pytest tests/test_ref_signature_gpu.py -m gpu -q -s 2>&1 | grep "shim timing" > $O/shim_timing.txt
python tools/gpu_latency.py > $O/host_api_latency.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O
