# attention on eight waves of 16 queries (option attn_waves8): bit-identity test, then the step A/B (alternating, same box)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_f16.py -m gpu -q -x -k "eight_waves or attention" > gpurun_out/b14_tests.log 2>&1 || { tail -40 gpurun_out/b14_tests.log; exit 1; }
tail -2 gpurun_out/b14_tests.log
LEGS="--no-cpu-baseline --no-full-chain --no-f32 --no-x3 --no-train --no-refine --no-cond"
for i in 1 2 3; do
  DN_ATTN_WAVES8=0 python bench.py $LEGS --steps 200 --warmup 20 > gpurun_out/b14_a0_$i.json 2>/dev/null
  DN_ATTN_WAVES8=1 python bench.py $LEGS --steps 200 --warmup 20 > gpurun_out/b14_a1_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("a0_1","a1_1","a0_2","a1_2","a0_3","a1_3"):
    d=json.loads(open(f'gpurun_out/b14_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['value'],2))
PY
