# parity sweeps + LocalBA determinism at the round's last code change: gpurun_out/fuzz_r06/*.txt (last lines go to profiles/r06_fuzz/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fuzz_r06; mkdir -p $O; cd $R
python tools/gpu_lba_determinism.py 3000 2>&1 | tail -4 > $O/lba_determinism.txt
python tools/gpu_fuzz_rest.py 100000 ${FUZZ_S:-150} 2>&1 | tail -3 > $O/rest.txt
python tools/gpu_fuzz_keyframes.py 100000 ${FUZZ_S:-150} 2>&1 | tail -3 > $O/keyframes.txt
python tools/gpu_fuzz_extractor.py 100000 ${FUZZ_S:-150} 2>&1 | tail -3 > $O/extractor.txt
python tools/gpu_fuzz_matcher.py 100000 ${FUZZ_S:-150} 2>&1 | tail -3 > $O/matcher.txt
python tools/gpu_fuzz_more.py 100000 ${FUZZ_S:-150} 2>&1 | tail -3 > $O/more.txt
tail -n 3 $O/*.txt
