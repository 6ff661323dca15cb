# how much does a batch whose padded length is not a multiple of 256 (sequence starts inside tiles: the taps' shared staged rows fall back
# to shifted copies there) lose?  frame-steps/s at T = 512, 437, 384
set -e
cd $GRAFT_REPO_ROOT
LEGS="--no-cpu-baseline --no-full-chain --no-f32 --no-x3 --no-train --no-refine --no-cond --steps 100 --warmup 10"
for T in 512 437 384 300; do
  python bench.py $LEGS --frames $T > gpurun_out/b19_T$T.json 2>/dev/null
  DN_FAT_HALO=0 DN_BIG_HALO=0 python bench.py $LEGS --frames $T > gpurun_out/b19_T${T}_nohalo.json 2>/dev/null
done
python - <<'PY'
import json
for T in (512,437,384,300):
    for sfx in ("","_nohalo"):
        d=json.loads(open(f'gpurun_out/b19_T{T}{sfx}.json').read().strip().splitlines()[-1]); print(T, sfx, round(d['value'],2), round(d['value']*T*32/1e6,3), 'Mframe-steps/s')
PY
