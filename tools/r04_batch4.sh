set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_f16.py tests/test_hip_engine.py -m gpu -q -k "every_tile or two_workgroup" > gpurun_out/b4_tests.log 2>&1 || true
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/b4_tests.log | head -20
F="--no-cpu-baseline --no-full-chain --no-x3 --no-train --no-f32 --no-refine --no-cond"
for i in 1 2; do
  python bench.py $F > gpurun_out/b4_a$i.json 2>/dev/null
  DN_MID2=1 python bench.py $F > gpurun_out/b4_b$i.json 2>/dev/null
  DN_MID2=2 python bench.py $F > gpurun_out/b4_c$i.json 2>/dev/null
  DN_MID2=3 python bench.py $F > gpurun_out/b4_d$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("a1","b1","c1","d1","a2","b2","c2","d2"):
    try:
        d=json.loads(open(f'gpurun_out/b4_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['value'],2))
    except Exception as e: print(n, 'failed', e)
PY
