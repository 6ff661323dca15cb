# Where the time of the short-K contractions on the 256 x 256 tile goes: diagnostic builds of gemm.hip on the GPU box
# (bit 16 = no epilogue, 2 = no MFMA, 1 = no staging DMA after the prologue, 4 = no LDS fragment reads).
set -e
cd $GRAFT_REPO_ROOT
for abl in 0 16 2 1 18; do
  touch diffnorm_amd/csrc/gemm.hip
  make -C diffnorm_amd/csrc EXTRA=-DDN_GEMM_ABL=$abl > gpurun_out/abl_build_$abl.log 2>&1
  echo "== DN_GEMM_ABL=$abl"
  timeout -k 10 120 python tools/gemm_bench.py bf16 shortk 2>&1 | grep -v amdgpu.ids
done
