set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_train_ops.py -m gpu -q -x > gpurun_out/b13_tests.log 2>&1 || { tail -40 gpurun_out/b13_tests.log; exit 1; }
tail -2 gpurun_out/b13_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/b13_prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 8 --warmup 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_trace.py $(find gpurun_out/b13_prof -name "*kernel_trace.csv" | head -1) > gpurun_out/b13_per_shape.txt
find gpurun_out/b13_prof -name "*.csv" -size +1M -delete
grep "rmsnorm_bwd\|attn_bwd_dq" gpurun_out/b13_per_shape.txt | cut -c1-150
