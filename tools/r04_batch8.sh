set -e
cd $GRAFT_REPO_ROOT
python tools/cond_ab.py f16 > gpurun_out/b8_cond_ab.txt 2>gpurun_out/b8_err.txt
cat gpurun_out/b8_cond_ab.txt
cd /tmp && export TMPDIR=/tmp && export COND_AB_ONLY=default
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/b8_prof -o cond -- python3 $GRAFT_REPO_ROOT/tools/cond_ab.py f16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/b8_prof | head
