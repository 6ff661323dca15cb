set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_f16.py tests/test_hip_train.py -m gpu -q -k "every_tile or fullsize_diffusion or bench_shape" > gpurun_out/b2_tests.log 2>&1 || true
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/b2_tests.log | head -20
F="--no-cpu-baseline --no-full-chain --no-x3 --no-train --no-f32 --no-refine --no-cond"
for i in 1 2; do
  python bench.py $F > gpurun_out/b2_a$i.json 2>/dev/null
  DN_QKV_192=1 python bench.py $F > gpurun_out/b2_b$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("a1","b1","a2","b2"):
    d=json.loads(open(f'gpurun_out/b2_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['value'],2), round(d['roofline']['avg_launch_ms']*1e3,1))
PY
python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b2_train_diff.json 2>/dev/null
python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b2_train_vae.json 2>/dev/null
python - <<'PY'
import json
for n in ("train_diff","train_vae"):
    d=json.loads(open(f'gpurun_out/b2_{n}.json').read().strip().splitlines()[-1]); print(n, d['value'], d['ms_per_step'])
PY
