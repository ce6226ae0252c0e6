# where a keyframe job's wall time goes: per-call walls from the native runner + the kernel timeline of two steps
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kf_trace; rm -rf $O; mkdir -p $O
python $R/bench.py --no-extra --no-cpu-baseline --no-verify --steps 40 --warmup 6 $BENCH_ARGS 2>$O/err1.txt | tail -1 > $O/line.json
python - <<PY
import json
d = json.loads(open("$O/line.json").read())
print(d["value"], d["ms_per_step"]); print(json.dumps(d["extra"]["timed_steps"], indent=0))
PY
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $R/bench.py --no-extra --no-cpu-baseline --no-verify --steps 8 --warmup 4 $BENCH_ARGS > $O/line_prof.json 2> $O/err2.txt
python $R/tools/kf_timeline.py $O/prof > $O/timeline.txt 2>&1
find $O/prof -name "*.csv" -size +20M -delete
tail -5 $O/timeline.txt
