cd $GRAFT_REPO_ROOT
(time timeout 1200 python -m pytest tests/test_gpu_native.py -m gpu -x -q) > gpurun_out/r2e_native.log 2>&1
tail -n 25 gpurun_out/r2e_native.log
