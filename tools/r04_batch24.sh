# the weight-gradient stream at the lowest priority (option wgrad_prio), with and without the 192-column tile: training A/B
set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  DN_WGRAD_PRIO=0 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b24_v00_$i.json 2>/dev/null
  DN_WGRAD_PRIO=1 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b24_v10_$i.json 2>/dev/null
  DN_WGRAD_PRIO=1 DN_WGRAD_K192=1 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b24_v11_$i.json 2>/dev/null
done
for i in 1 2; do
  DN_WGRAD_PRIO=0 python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b24_d0_$i.json 2>/dev/null
  DN_WGRAD_PRIO=1 python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b24_d1_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("v00_1","v10_1","v11_1","v00_2","v10_2","v11_2","v00_3","v10_3","v11_3","d0_1","d1_1","d0_2","d1_2"):
    d=json.loads(open(f'gpurun_out/b24_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2))
PY
