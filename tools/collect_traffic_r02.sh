# PMC passes (counters only, one per run): fabric traffic of one denoising step, and of the training bench's roofline kernel
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
F="--no-split --no-cpu-baseline --no-full-chain --no-f32"
for n in 10 30; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/steptraffic2/${c}_$n -o p -- python3 $R/bench.py $F --steps $n > $R/gpurun_out/steptraffic2_${c}_$n.log 2>&1
  echo done $c $n
done; done
python tools/step_traffic.py gpurun_out/steptraffic2 10 30 > gpurun_out/r02_step_traffic.json
cat gpurun_out/r02_step_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traintraffic/$c -o p -- python3 $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/traintraffic_$c.log 2>&1
  echo done train $c
done
python tools/pmc_summary.py "conv_gemm_big_kernel<dn::BF16, 4>@32768x3" gpurun_out/traintraffic/* > gpurun_out/r02_pmc_train_wgrad.json
cat gpurun_out/r02_pmc_train_wgrad.json
find gpurun_out/steptraffic2 gpurun_out/traintraffic -name "*.csv" -size +2M -delete
