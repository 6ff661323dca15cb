# split-operand attention on eight waves: bit-identity test, then the bf16x3 chain A/B
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_f16.py tests/test_hip_bf16x3.py -m gpu -q -x > gpurun_out/b18_tests.log 2>&1 || { tail -40 gpurun_out/b18_tests.log; exit 1; }
tail -2 gpurun_out/b18_tests.log
LEGS="--no-cpu-baseline --no-full-chain --no-f32 --no-x3 --no-train --no-refine --no-cond --dtype bf16x3"
for i in 1 2; do
  DN_ATTN_WAVES8=0 python bench.py $LEGS --steps 100 --warmup 10 > gpurun_out/b18_a0_$i.json 2>/dev/null
  DN_ATTN_WAVES8=1 python bench.py $LEGS --steps 100 --warmup 10 > gpurun_out/b18_a1_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("a0_1","a1_1","a0_2","a1_2"):
    d=json.loads(open(f'gpurun_out/b18_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['value'],2))
PY
