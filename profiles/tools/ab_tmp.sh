cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_dp.py -x -q 2>&1 | tail -15
python3 profiles/tools/dp_overhead.py 2>&1 | grep "ms/step"
