# Round-4 evidence, one gpurun call: the default bench line (f16 headline), rocprofv3 per-shape summaries (f16 one stream / two streams,
# bf16 one stream, bf16x3, both training losses) and the PMC passes (counters only, one counter per run) for the step traffic and the
# dominant kernel of the f16 and bf16x3 chains.  Everything lands under gpurun_out/r04/; tools/publish_profiles_r04.py copies the
# summaries to profiles/.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
O=gpurun_out/r04
rm -rf $O && mkdir -p $O
BID=$(python tools/build_id.py)
echo "build $BID" > $O/build_id.txt
python bench.py > $O/bench_line.json 2> $O/bench_line.err
echo "bench line done"
LEGS="--no-cpu-baseline --no-full-chain --no-f32 --no-x3 --no-train --no-refine --no-cond"
prof() {  # name, bench flags
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$1 -o bench -- python3 $R/bench.py $2 > $O/rocprof_$1.json 2> $O/rocprof_$1.err
  python tools/summarize_trace.py $(find $O/prof_$1 -name "*kernel_trace.csv" | head -1) > $O/per_shape_$1.txt
  python tools/step_sequence.py $(find $O/prof_$1 -name "*kernel_trace.csv" | head -1) > $O/step_sequence_$1.txt 2>/dev/null || true
  cp $(find $O/prof_$1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$1.csv
  find $O/prof_$1 -name "*.csv" -size +1M -delete
  echo "profile $1 done"
}
prof f16_one_stream "$LEGS --steps 30 --no-split"
prof f16_split_streams "$LEGS --steps 30"
prof bf16_one_stream "$LEGS --steps 30 --no-split --dtype bf16"
prof bf16x3_one_stream "$LEGS --steps 20 --no-split --dtype bf16x3"
prof train_vae "--mode train --steps 8 --warmup 4"
prof train_diffusion "--mode train --train-loss diffusion --max-tokens 12000 --steps 5 --warmup 4"
for dt in f16 bf16x3; do for n in 10 30; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$dt/${c}_$n -o p -- python3 $R/bench.py $LEGS --no-split --dtype $dt --steps $n > $O/pmc_${dt}_${c}_$n.log 2>&1
  echo "pmc $dt $c $n done"
done; done
  python tools/step_traffic.py $O/pmc_$dt 10 30 > $O/step_traffic_$dt.json
done
python tools/pmc_summary.py "conv_gemm_fat_kernel" $O/pmc_f16/FETCH_SIZE_30 $O/pmc_f16/WRITE_SIZE_30 > $O/pmc_ffn_conv_f16.json
python tools/pmc_summary.py "conv_gemm_big_kernel<dn::BF16X3, 0, false, 256@196608" $O/pmc_bf16x3/FETCH_SIZE_30 $O/pmc_bf16x3/WRITE_SIZE_30 > $O/pmc_ffn_conv_bf16x3.json
python tools/step_traffic_by_kernel.py $O/pmc_f16 > $O/step_traffic_by_kernel_f16.txt 2>/dev/null || true
python tools/step_traffic_by_kernel.py $O/pmc_bf16x3 > $O/step_traffic_by_kernel_bf16x3.txt 2>/dev/null || true
# the weight-gradient kernel of the training updates (roofline.traffic of the train legs)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_train/$c -o p -- python3 $R/bench.py --mode train --steps 6 --warmup 3 > $O/pmc_train_$c.log 2>&1
  echo "pmc train $c done"
done
python tools/pmc_summary.py "wgrad_tn_kernel@98304x1" $O/pmc_train/FETCH_SIZE $O/pmc_train/WRITE_SIZE > $O/pmc_wgrad_vae.json 2>/dev/null || true
find $O -name "*.csv" -size +2M -delete
cat $O/step_traffic_f16.json $O/step_traffic_bf16x3.json $O/pmc_ffn_conv_f16.json $O/pmc_ffn_conv_bf16x3.json
