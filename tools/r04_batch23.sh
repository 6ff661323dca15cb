# weight-gradient kernel on 192 k-columns per tile where that fills the chip (VAE FFN conv: 192 -> 256 workgroups): tests, A/B
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_train_ops.py tests/test_hip_train.py -m gpu -q -x > gpurun_out/b23_tests.log 2>&1 || { tail -40 gpurun_out/b23_tests.log; exit 1; }
tail -2 gpurun_out/b23_tests.log
for i in 1 2 3; do
  DN_WGRAD_K192=0 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b23_v0_$i.json 2>/dev/null
  python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b23_v1_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("v0_1","v1_1","v0_2","v1_2","v0_3","v1_3"):
    d=json.loads(open(f'gpurun_out/b23_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms']*1e3,1), round(d['roofline']['frac'],3))
PY
