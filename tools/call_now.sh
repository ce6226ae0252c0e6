# scratch: the command of the last gpurun call of the round (the full gpu suite + smoke at the final commit)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.log 2>&1; tail -2 gpurun_out/final/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
