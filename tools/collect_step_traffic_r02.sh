set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
F="--no-split --no-cpu-baseline --no-full-chain --no-f32"
rm -rf gpurun_out/steptraffic2
for n in 10 30; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/steptraffic2/${c}_$n -o p -- python3 $R/bench.py $F --steps $n > $R/gpurun_out/steptraffic2_${c}_$n.log 2>&1
done; done
python tools/step_traffic.py gpurun_out/steptraffic2 10 30 > gpurun_out/r02_step_traffic.json
cat gpurun_out/r02_step_traffic.json
find gpurun_out/steptraffic2 -name "*.csv" -size +2M -delete
