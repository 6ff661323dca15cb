set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
for n in 10 30; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/steptraffic/${c}_$n -o p -- python3 $R/bench.py --no-split --no-cpu-baseline --steps $n > $R/gpurun_out/steptraffic_${c}_$n.log 2>&1
  echo done $c $n
done; done
python tools/step_traffic.py gpurun_out/steptraffic 10 30 > gpurun_out/step_traffic.json
cat gpurun_out/step_traffic.json
