set -e
cd $GRAFT_REPO_ROOT
python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b10_v.json 2>/dev/null
python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b10_d.json 2>/dev/null
python - <<'PY'
import json
for n in ("v","d"):
    d=json.loads(open(f'gpurun_out/b10_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_update'],2))
PY
