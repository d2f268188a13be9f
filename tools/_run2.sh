set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests/test_gpu_filter_and_configs.py -m gpu -x -q) > gpurun_out/r2d_newtests.log 2>&1
echo "newtests rc=$?" >> gpurun_out/r2d_newtests.log
(time timeout 1800 python -m pytest tests -m gpu -q -x) > gpurun_out/r2d_alltests.log 2>&1
echo "alltests rc=$?" >> gpurun_out/r2d_alltests.log
timeout 600 python bench.py > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err
echo "bench rc=$?" >> gpurun_out/r2d_bench.err
timeout 900 python tools/bench_configs.py > gpurun_out/r2d_configs.jsonl 2> gpurun_out/r2d_configs.err
timeout 300 python tools/tune.py --gib 8 --needles 1,2,4,8,16,32,128,2000 --variants 0 > gpurun_out/r2d_tune8.jsonl 2>&1
timeout 1500 bash tools/profile_kernels.sh kp2 > gpurun_out/r2d_profile.log 2>&1
tail -n 5 gpurun_out/r2d_newtests.log
tail -n 5 gpurun_out/r2d_alltests.log
cut -c1-300 gpurun_out/r2d_bench.json
