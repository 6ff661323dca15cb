# per-kernel fabric traffic and rate of the training bench (two PMC passes); optional env passes through (e.g. DN_GEMM_BAND=1)
set -e
R=$GRAFT_REPO_ROOT
OUT=${1:-traintraffic}
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/$OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/$OUT/$c -o p -- python3 $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${OUT}_$c.log 2>&1
done
python tools/traffic_rate_by_kernel.py gpurun_out/$OUT > gpurun_out/${OUT}_by_kernel.txt
head -14 gpurun_out/${OUT}_by_kernel.txt
