# same-box A/B of an environment knob on the sampling bench: tools/ab_bench.sh "<bench flags>" VAR=a VAR=b [reps]
F="$1"; A="$2"; B="$3"; N=${4:-2}
for i in $(seq $N); do for kv in "$A" "$B"; do
  v=$(env $kv python bench.py $F --no-full-chain --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['value'])")
  echo "$kv  $v steps/s"
done; done
