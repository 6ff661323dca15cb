set -e
cd $GRAFT_REPO_ROOT
S=$(date +%s)
python bench.py > gpurun_out/b5_bench_line.json 2> gpurun_out/b5_bench_line.err || { tail -30 gpurun_out/b5_bench_line.err; exit 1; }
echo "default bench wall: $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b5_bench_line.json').read().strip().splitlines()[-1])
keys=['value','dtype','ms_per_step','step_mfma_frac','bf16_steps_per_s','bf16x3_steps_per_s','ddpm_steps_per_s','backend','rccl_ranks','f32_steps_per_s']
print({k:d.get(k) for k in keys})
print('roofline', {k:d['roofline'][k] for k in ('frac','avg_launch_ms','traffic')})
print('full_chain', d.get('full_chain',{}).get('steps_per_s_incl_setup_and_vae'), 'f16_vs_f32', d.get('f16_vs_f32_after_full_chain'))
print('train', {k:(v['ms_per_update'], v['samples_per_s'], v['step_mfma_frac']) for k,v in d.get('train',{}).items()})
print('refine', {k:d['refine'][k] for k in ('ms_per_iteration','ms_per_iteration_host_stepped','speech_encoder_ms_per_batch')})
print('cond', d['cond']['ms_per_guided_step'], 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('gpu_over_cpu'))
PY
