# Round 4: kernel traces of the two training updates (per-shape summaries + where the device copies sit).  -> gpurun_out/r04t/
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
O=gpurun_out/r04t
rm -rf $O && mkdir -p $O
python tools/build_id.py > $O/build_id.txt
prof() {  # name, bench flags
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$1 -o bench -- python3 $R/bench.py $2 > $O/rocprof_$1.json 2> $O/rocprof_$1.err
  T=$(find $O/prof_$1 -name "*kernel_trace.csv" | head -1)
  python tools/summarize_trace.py $T > $O/per_shape_$1.txt
  python tools/trace_neighbors.py $T copyBuffer 80 > $O/copies_$1.txt || true
  python tools/trace_neighbors.py $T colsum_partial 40 > $O/colsum_$1.txt || true
  find $O/prof_$1 -name "*.csv" -size +1M -delete
  echo "profile $1 done"; tail -c 600 $O/rocprof_$1.json
}
prof train_vae "--mode train --steps 8 --warmup 4"
prof train_diffusion "--mode train --train-loss diffusion --max-tokens 12000 --steps 5 --warmup 4"
