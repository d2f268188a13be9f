set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$PWD/sliceslice-rs_amd/csrc
RT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.tsan-x86_64.so)
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -pthread -fsanitize=thread -fno-gpu-sanitize -shared-libsan -I include tests/native/host_stress_test.cpp -o /tmp/hst_tsan -L $C -lsliceslice_hip_tsan -Wl,-rpath,$C -Wl,-rpath,$(dirname $RT) > gpurun_out/tsan_build.log 2>&1
echo "build rc=$?" >> gpurun_out/tsan_build.log
(time TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:history_size=4" timeout 900 /tmp/hst_tsan 48) > gpurun_out/tsan_run.log 2>&1
echo "run rc=$?" >> gpurun_out/tsan_run.log
grep -c "WARNING: ThreadSanitizer" gpurun_out/tsan_run.log
tail -5 gpurun_out/tsan_run.log
timeout 400 python tools/fuzz_gpu.py 240 21 > gpurun_out/fuzz_small.json 2> gpurun_out/fuzz_small.err; echo "fuzz rc=$?"
timeout 300 python tools/fuzz_gpu.py 150 22 6 > gpurun_out/fuzz_big.json 2> gpurun_out/fuzz_big.err; echo "fuzz big rc=$?"
cat gpurun_out/fuzz_small.json gpurun_out/fuzz_big.json
