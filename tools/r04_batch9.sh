# colsum / attn_delta rewrites + 256 x 192 tile for lone launches: tests, then training A/B (tile_192 on/off), then traces
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_f16.py tests/test_hip_train.py tests/test_hip_ops.py tests/test_hip_train_ops.py -m gpu -q -x > gpurun_out/b9_tests.log 2>&1 || { tail -40 gpurun_out/b9_tests.log; exit 1; }
tail -2 gpurun_out/b9_tests.log
for i in 1 2; do
  DN_TILE_192=0 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b9_v0_$i.json 2>/dev/null
  python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b9_v1_$i.json 2>/dev/null
  DN_TILE_192=0 python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b9_d0_$i.json 2>/dev/null
  python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b9_d1_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("v0_1","v1_1","v0_2","v1_2","d0_1","d1_1","d0_2","d1_2"):
    d=json.loads(open(f'gpurun_out/b9_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2))
PY
bash tools/prof_train_r04.sh > gpurun_out/b9_prof.log 2>&1
tail -5 gpurun_out/b9_prof.log
