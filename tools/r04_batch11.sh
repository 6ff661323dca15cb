# fused GEGLU (+ kept pre-activation) in the training forward, rmsnorm backward with a row in flight: tests, A/B, traces
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_f16.py tests/test_hip_train.py tests/test_hip_train_ops.py tests/test_hip_engine.py -m gpu -q -x > gpurun_out/b11_tests.log 2>&1 || { tail -40 gpurun_out/b11_tests.log; exit 1; }
tail -2 gpurun_out/b11_tests.log
for i in 1 2; do
  DN_FUSED_GEGLU=0 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b11_v0_$i.json 2>/dev/null
  python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b11_v1_$i.json 2>/dev/null
  DN_FUSED_GEGLU=0 python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b11_d0_$i.json 2>/dev/null
  python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b11_d1_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("v0_1","v1_1","v0_2","v1_2","d0_1","d1_1","d0_2","d1_2"):
    d=json.loads(open(f'gpurun_out/b11_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2))
PY
