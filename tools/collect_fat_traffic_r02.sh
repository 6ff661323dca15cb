# fabric traffic of the dominant kernel (FFN conv on the 256x352 tile, K-blocked operands as the engine runs it): taps innermost (default) and term-outer
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
for mode in inner outer; do
  if [ $mode = outer ]; then export DN_TAPS_INNER=0; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fat_$mode/$c -o p -- python3 $R/tools/gemm_bench.py bf16 ffn > $R/gpurun_out/pmc_fat_${mode}_$c.log 2>&1
  done
  python tools/pmc_summary.py conv_gemm_fat_kernel gpurun_out/pmc_fat_$mode/* > gpurun_out/r02_pmc_fat_$mode.json
  cat gpurun_out/r02_pmc_fat_$mode.json
done
find gpurun_out/pmc_fat_inner gpurun_out/pmc_fat_outer -name "*.csv" -size +2M -delete
