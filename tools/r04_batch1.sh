set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_mirror.py tests/test_hip_refine.py tests/test_hip_train.py tests/test_hip_fullsize.py -m gpu -q -s -k "gaussian_diffusion_tail or refine or bench_shape or fullsize_vae or fullsize_diffusion or conditional or level1" > gpurun_out/b1_tests.log 2>&1 || true
grep -E "^(FAILED|ERROR)|passed|failed|cosines|worst relative" gpurun_out/b1_tests.log | head -40
timeout -k 10 120 python tools/cond_gemm_bench.py > gpurun_out/b1_cond_gemm.log 2>&1 || true
cat gpurun_out/b1_cond_gemm.log | tail -6
timeout -k 10 300 python bench.py --no-cpu-baseline --no-full-chain --no-x3 --no-train --no-f32 > gpurun_out/b1_bench.json 2> gpurun_out/b1_bench.err || { tail -20 gpurun_out/b1_bench.err; }
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/b1_bench.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','dtype','ms_per_step','bf16_steps_per_s') if k in d}); print(d.get('refine')); print(d.get('cond'))
except Exception as e: print('bench parse', e)
PY
bash tools/prof_train_r04.sh > gpurun_out/b1_prof.log 2>&1 || tail -20 gpurun_out/b1_prof.log
head -12 gpurun_out/r04t/copies_train_diffusion.txt; head -6 gpurun_out/r04t/copies_train_vae.txt
