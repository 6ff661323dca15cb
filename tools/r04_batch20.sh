# conditioning projection's data gradient streamed from the row-major master matrix: operator test, training tests, diffusion A/B
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_train_ops.py tests/test_hip_train.py -m gpu -q -x -k "rows_times or diffusion" > gpurun_out/b20_tests.log 2>&1 || { tail -40 gpurun_out/b20_tests.log; exit 1; }
tail -2 gpurun_out/b20_tests.log
for i in 1 2 3; do
  DN_COND_STREAM=0 python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b20_d0_$i.json 2>/dev/null
  python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b20_d1_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("d0_1","d1_1","d0_2","d1_2","d0_3","d1_3"):
    d=json.loads(open(f'gpurun_out/b20_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2))
PY
