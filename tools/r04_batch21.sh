# rehearsal of the N = 2 control flow of bench.py (two ranks sharing the one GPU over gloo): sampling line and both training lines
set -e
cd $GRAFT_REPO_ROOT
export DN_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/b21_s.json 2> gpurun_out/b21_s.err || { tail -20 gpurun_out/b21_s.err; exit 1; }
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --mode train --steps 4 --warmup 2 > gpurun_out/b21_v.json 2> gpurun_out/b21_v.err || { tail -20 gpurun_out/b21_v.err; exit 1; }
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --mode train --train-loss diffusion --max-tokens 12000 --steps 3 --warmup 2 > gpurun_out/b21_d.json 2> gpurun_out/b21_d.err || { tail -20 gpurun_out/b21_d.err; exit 1; }
python - <<'PY'
import json
for n in ("s","v","d"):
    d=json.loads(open(f'gpurun_out/b21_{n}.json').read().strip().splitlines()[-1]); print(n, d['n_gpus'], d['backend'], d['rccl_ranks'], d['process_group_ranks'], round(d['value'],2), d['unit'], d.get('all_reduce_ms_per_update'))
PY
