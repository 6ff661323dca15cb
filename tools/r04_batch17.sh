# eight-wave attention (forward, dK/dV, dQ up to 64 dims): tests, training A/B, kernel times
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_f16.py tests/test_hip_ops.py tests/test_hip_train_ops.py tests/test_hip_train.py -m gpu -q -x > gpurun_out/b17_tests.log 2>&1 || { tail -40 gpurun_out/b17_tests.log; exit 1; }
tail -2 gpurun_out/b17_tests.log
for i in 1 2; do
  DN_ATTN_WAVES8=0 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b17_v0_$i.json 2>/dev/null
  python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/b17_v1_$i.json 2>/dev/null
  DN_ATTN_WAVES8=0 python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b17_d0_$i.json 2>/dev/null
  python bench.py --mode train --train-loss diffusion --max-tokens 12000 --steps 6 --warmup 3 > gpurun_out/b17_d1_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("v0_1","v1_1","v0_2","v1_2","d0_1","d1_1","d0_2","d1_2"):
    d=json.loads(open(f'gpurun_out/b17_{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2))
PY
cd /tmp && export TMPDIR=/tmp
for k in vae diffusion; do
  F="--mode train --steps 6 --warmup 4"; [ $k = diffusion ] && F="--mode train --train-loss diffusion --max-tokens 12000 --steps 5 --warmup 4"
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/b17_prof_$k -o bench -- python3 $GRAFT_REPO_ROOT/bench.py $F > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/summarize_trace.py $(find $GRAFT_REPO_ROOT/gpurun_out/b17_prof_$k -name "*kernel_trace.csv" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/b17_per_shape_$k.txt
  find $GRAFT_REPO_ROOT/gpurun_out/b17_prof_$k -name "*.csv" -size +1M -delete
  grep "attn_" $GRAFT_REPO_ROOT/gpurun_out/b17_per_shape_$k.txt | cut -c1-150
done
