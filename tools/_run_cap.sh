set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/capture_profiles.sh r02cap > gpurun_out/r02cap.log 2>&1
bash tools/profile_kernels.sh kp3 > gpurun_out/r02cap_kernels.log 2>&1
GD=tests/golden/data
tools/native_bench latency 2000 > gpurun_out/r02cap_latency.json 2>> gpurun_out/r02cap.log
SLICESLICE_SPIN_WAIT=0 tools/native_bench latency 2000 > gpurun_out/r02cap_latency_nospin.json 2>> gpurun_out/r02cap.log
tools/native_bench config1 $GD/i386.txt $GD/words.txt 5 > gpurun_out/r02cap_config1.json 2>> gpurun_out/r02cap.log
tools/native_bench headline 64 20 > gpurun_out/r02cap_native_headline.json 2>> gpurun_out/r02cap.log
python tools/tune.py --gib 0.25 --needles 16 --variants 0 > gpurun_out/r02cap_size.jsonl 2>/dev/null
for g in 1 2 4 8 16 32 64; do python tools/tune.py --gib $g --needles 1,16 --variants 0 >> gpurun_out/r02cap_size.jsonl 2>/dev/null; done
python tools/host_path_bench.py > gpurun_out/r02cap_host_path.jsonl 2>> gpurun_out/r02cap.log
cut -c1-200 gpurun_out/r02cap_bench.json; tail -3 gpurun_out/r02cap.log
