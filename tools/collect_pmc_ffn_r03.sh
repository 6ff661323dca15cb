# PMC comparison of the FFN causal conv on the 256x352 tile with a shifted copy of the rows per tap (DN_FAT_HALO=0) and with one
# shared staged copy (DN_FAT_HALO=1): counters only (one small group per pass, no trace domains besides the kernel trace),
# isolated launches of tools/gemm_bench.py bf16 ffn.  -> gpurun_out/pmc_ffn/{halo0,halo1}.json
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
O=gpurun_out/pmc_ffn
rm -rf $O && mkdir -p $O
python tools/build_id.py > $O/build_id.txt
for h in 0 1; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" "TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    DN_FAT_HALO=$h rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$O/h$h/g$i -o p -- python3 $R/tools/gemm_bench.py bf16 ffn > $O/h${h}_g$i.log 2>&1 || echo "group $i ($grp) failed"
    echo "halo=$h group $i done"
  done
  python tools/pmc_summary.py "conv_gemm_fat_kernel" $O/h$h/g* > $O/halo$h.json
done
find $O -name "*.csv" -size +1M -delete
cat $O/halo0.json $O/halo1.json
