set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_hip_f16.py tests/test_hip_bf16_model.py tests/test_hip_engine.py tests/test_hip_fullsize.py tests/test_hip_refine.py -m gpu -q -s -k "f16" > gpurun_out/f16_tests.log 2>&1 || true
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/f16_tests.log | head -40
