set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
python bench.py > gpurun_out/r01_bench_line.json 2> gpurun_out/r01_bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01prof -o bench -- python3 $R/bench.py --no-split > gpurun_out/r01_bench_under_rocprof.json 2> gpurun_out/r01_bench_under_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_kb/$c -o p -- python3 $R/tools/gemm_bench.py bf16 ffn > $R/gpurun_out/pmc_kb_$c.log 2>&1
done
python tools/pmc_summary.py conv_gemm_fat_kernel gpurun_out/pmc_kb/* > gpurun_out/pmc_kb_fat.json
find gpurun_out/r01prof -name "*.csv" | head
cat gpurun_out/pmc_kb_fat.json
