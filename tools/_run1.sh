set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests/test_gpu_filter_and_configs.py tests/test_gpu_sharded.py tests/test_gpu_native.py -m gpu -x -q) > gpurun_out/r2a_newtests.log 2>&1
echo "newtests rc=$?" >> gpurun_out/r2a_newtests.log
(time timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/r2a_alltests.log 2>&1
echo "alltests rc=$?" >> gpurun_out/r2a_alltests.log
timeout 600 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
echo "bench rc=$?" >> gpurun_out/r2a_bench.err
timeout 900 python tools/bench_configs.py > gpurun_out/r2a_configs.jsonl 2> gpurun_out/r2a_configs.err
tail -5 gpurun_out/r2a_newtests.log gpurun_out/r2a_alltests.log
cat gpurun_out/r2a_bench.json | cut -c1-600
