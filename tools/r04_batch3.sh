set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_refine.py -m gpu -q -s > gpurun_out/b3_tests.log 2>&1 || true
grep -E "^(FAILED|ERROR)|passed|failed|max abs err|^E  " gpurun_out/b3_tests.log | head -40
