# rocprofv3 per-kernel traces of the sampling bench (default = two forked half-batch streams; --no-split = one stream, clean per-kernel times)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
F="--steps 30 --warmup 5 --no-full-chain --no-f32 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02prof_split -o bench -- python3 $R/bench.py $F > gpurun_out/r02_rocprof_split.json 2> gpurun_out/r02_rocprof_split.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02prof_nosplit -o bench -- python3 $R/bench.py $F --no-split > gpurun_out/r02_rocprof_nosplit.json 2> gpurun_out/r02_rocprof_nosplit.err
python tools/summarize_trace.py $(find gpurun_out/r02prof_split -name "*kernel_trace.csv" | head -1) > gpurun_out/r02_per_shape_split.txt
python tools/summarize_trace.py $(find gpurun_out/r02prof_nosplit -name "*kernel_trace.csv" | head -1) > gpurun_out/r02_per_shape_nosplit.txt
cp $(find gpurun_out/r02prof_split -name "*kernel_stats.csv" | head -1) gpurun_out/r02_kernel_stats_split.csv
cp $(find gpurun_out/r02prof_nosplit -name "*kernel_stats.csv" | head -1) gpurun_out/r02_kernel_stats_nosplit.csv
find gpurun_out/r02prof_split gpurun_out/r02prof_nosplit -name "*.csv" -size +1M -delete
head -14 gpurun_out/r02_per_shape_split.txt; head -14 gpurun_out/r02_per_shape_nosplit.txt
