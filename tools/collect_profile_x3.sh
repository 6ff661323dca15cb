# rocprofv3 per-kernel trace of the sampling bench in bf16x3 mode (one stream: clean per-kernel times)
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03_x3}
cd /tmp && export TMPDIR=/tmp
cd $R
F="--dtype bf16x3 --steps 20 --warmup 5 --no-full-chain --no-f32 --no-cpu-baseline --no-split"
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o bench -- python3 $R/bench.py $F > gpurun_out/${TAG}_rocprof.json 2> gpurun_out/${TAG}_rocprof.err
python tools/summarize_trace.py $(find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" | head -1) > gpurun_out/${TAG}_per_shape.txt
cp $(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
find gpurun_out/${TAG}_prof -name "*.csv" -size +1M -delete
head -24 gpurun_out/${TAG}_per_shape.txt
