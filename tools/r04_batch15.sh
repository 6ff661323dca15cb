set -e
cd $GRAFT_REPO_ROOT
LEGS="--no-cpu-baseline --no-full-chain --no-f32 --no-x3 --no-train --no-cond --steps 20 --warmup 5"
DN_TILE_192=0 python bench.py $LEGS > gpurun_out/b15_t0.json 2>/dev/null
python bench.py $LEGS > gpurun_out/b15_t1.json 2>/dev/null
python - <<'PY'
import json
for n in ("t0","t1"):
    d=json.loads(open(f'gpurun_out/b15_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['value'],2), d['refine']['speech_encoder_ms_per_batch'], d['refine']['ms_per_iteration'])
PY
