# the driver's GPU tier, run from the builder's side: the whole -m gpu suite in one process + the default bench line
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=60 > gpurun_out/gpu_suite.log 2>&1 || { tail -60 gpurun_out/gpu_suite.log; exit 1; }
tail -3 gpurun_out/gpu_suite.log
