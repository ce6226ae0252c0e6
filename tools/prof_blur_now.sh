cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
AOS2_DESC_BLUR=level rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/blurB -- python $R/tools/gpu_desc_blur_ab.py > /dev/null 2>&1; python $R/tools/kstats.py $R/gpurun_out/blurB 8
find $R/gpurun_out/blurB -name "*.db" -delete; find $R/gpurun_out/blurB -name "*trace.csv" -delete
