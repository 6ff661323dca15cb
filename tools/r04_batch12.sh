set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_train_ops.py tests/test_hip_ops.py -m gpu -q -x -k "attention or attn" > gpurun_out/b12_tests.log 2>&1 || { tail -40 gpurun_out/b12_tests.log; exit 1; }
tail -2 gpurun_out/b12_tests.log
for i in 1 2 3; do
  python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b12_v_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("v_1","v_2","v_3"):
    d=json.loads(open(f'gpurun_out/b12_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/b12_prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 8 --warmup 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_trace.py $(find gpurun_out/b12_prof -name "*kernel_trace.csv" | head -1) > gpurun_out/b12_per_shape.txt
find gpurun_out/b12_prof -name "*.csv" -size +1M -delete
grep "attn_" gpurun_out/b12_per_shape.txt | cut -c1-150
