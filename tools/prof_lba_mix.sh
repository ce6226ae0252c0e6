# kernel stats of one LocalBA batch per window mix (het / het26 / hom): gpurun_out/prof_lba_mix/<mix>_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_lba_mix
rm -rf $O; mkdir -p $O
for m in ${MIXES:-het het26 hom}; do
  LBA_MIX=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -- python $R/tools/gpu_lba_mix_prof.py > $O/$m.log 2>&1
  grep windows $O/$m.log
  python $R/tools/kstats.py $O/$m 14 | tee $O/${m}_kernel_stats.txt
  f=$(find $O/$m -name "*kernel_stats.csv" | head -1); cp $f $O/${m}_kernel_stats.csv
  find $O/$m -name "*.db" -delete; find $O/$m -name "*trace.csv" -delete
done
