# rocprofv3 per-(kernel, grid) summary of the training bench
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/r02prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02prof_train -o bench -- python3 $R/bench.py --mode train --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/r02_rocprof_train.json 2> gpurun_out/r02_rocprof_train.err
python tools/summarize_trace.py $(find gpurun_out/r02prof_train -name "*kernel_trace.csv" | head -1) > gpurun_out/r02_train_per_shape.txt
cp $(find gpurun_out/r02prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/r02_train_kernel_stats.csv
find gpurun_out/r02prof_train -name "*.csv" -size +1M -delete
head -42 gpurun_out/r02_train_per_shape.txt
