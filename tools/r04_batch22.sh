# tile-score factors for long multi-term contractions recalibrated (the VAE's FFN conv forward -> 256 x 256 with shared rows, backward-data
# -> 256 x 128): microbench, tests, training A/B (DN_GEMM_HEUR=5 = the rule before)
set -e
cd $GRAFT_REPO_ROOT
python tools/bwd_tiles.py > gpurun_out/b22_tiles.txt 2>/dev/null
grep "tile 0" gpurun_out/b22_tiles.txt
timeout -k 10 900 python -m pytest tests/test_hip_train.py tests/test_hip_ops.py -m gpu -q -x > gpurun_out/b22_tests.log 2>&1 || { tail -40 gpurun_out/b22_tests.log; exit 1; }
tail -2 gpurun_out/b22_tests.log
for i in 1 2 3; do
  DN_GEMM_HEUR=5 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b22_v0_$i.json 2>/dev/null
  python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b22_v1_$i.json 2>/dev/null
done
for i in 1 2; do
  DN_GEMM_HEUR=5 python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b22_d0_$i.json 2>/dev/null
  python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b22_d1_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("v0_1","v1_1","v0_2","v1_2","v0_3","v1_3","d0_1","d1_1","d0_2","d1_2"):
    d=json.loads(open(f'gpurun_out/b22_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2))
PY
