# conditional variant through the shared transformer path (split RMSNorm, K-blocked operands): parity tests + the cond leg
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_engine.py tests/test_hip_fullsize.py tests/test_hip_mirror.py -m gpu -q -x -k "cond or guided or prompt" > gpurun_out/b7_tests.log 2>&1 || { tail -40 gpurun_out/b7_tests.log; exit 1; }
tail -3 gpurun_out/b7_tests.log
python bench.py --no-cpu-baseline --no-x3 --no-train --no-refine --no-f32 --steps 20 --warmup 5 > gpurun_out/b7_bench.json 2>gpurun_out/b7_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b7_bench.json').read().strip().splitlines()[-1])
print(d['value'], json.dumps(d.get('cond') or {k:v for k,v in d.items() if 'cond' in k}))
PY
