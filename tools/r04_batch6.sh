set -e
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  DN_WGRAD_STAGES=4 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b6_v4_$i.json 2>/dev/null
  DN_WGRAD_STAGES=0 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b6_v0_$i.json 2>/dev/null
  DN_WGRAD_STAGES=4 python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b6_d4_$i.json 2>/dev/null
  DN_WGRAD_STAGES=0 python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b6_d0_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("v4_1","v0_1","v4_2","v0_2","d4_1","d0_1","d4_2","d0_2"):
    d=json.loads(open(f'gpurun_out/b6_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms']*1e3,1), round(d['roofline']['frac'],3))
PY
