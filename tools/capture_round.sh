#!/bin/bash
# Everything a round's profiles/<round>/ is made of, in one gpurun call (run from the repo root):
#   bash tools/capture_round.sh r04        -> gpurun_out/r04_* (capture_profiles.sh), gpurun_out/r04k_* (profile_kernels.sh),
#                                             gpurun_out/r04_extra/ (native_bench modes, service breakdown, fuzz campaigns, occupancy probe)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-cap}
X=$R/gpurun_out/${TAG}_extra
mkdir -p $X
L=$R/sliceslice-rs_amd/csrc
export LD_LIBRARY_PATH=$L:${LD_LIBRARY_PATH:-}
bash $R/tools/capture_profiles.sh $TAG > $X/capture.log 2>&1
bash $R/tools/profile_kernels.sh ${TAG}k > $X/kernels.log 2>&1
cd $R
tools/native_bench config1 tests/golden/data/i386.txt tests/golden/data/words.txt 5 2>/dev/null | grep '^{' > $X/native_config1.json
tools/native_bench latency 2000 2>/dev/null | grep '^{' > $X/native_latency.json
tools/native_bench sharded 8 50 2>/dev/null | grep '^{' > $X/native_sharded.json
for n in 2 4 8; do SLICESLICE_RCCL_LIB=$R/tests/native/libfake_rccl.so tools/native_bench ranks $n 8 200 2>/dev/null | grep '^{' >> $X/native_ranks.jsonl; done
for n in 2 8; do SLICESLICE_RCCL_LIB=$R/tests/native/libfake_rccl.so tools/native_bench ranks $n 0.0625 2000 2>/dev/null | grep '^{' >> $X/native_ranks.jsonl; done
SLICESLICE_HIP_LIB=$L/libsliceslice_hip_tuning.so SLICESLICE_SERVICE_DEBUG=1 python tools/service_breakdown.py > $X/service_breakdown.jsonl 2> $X/service_breakdown.err
SLICESLICE_HIP_LIB=$L/libsliceslice_hip_tuning.so SLICESLICE_SERVICE_DEBUG=1 SLICESLICE_SERVICE_HDP_FLUSH=0 python tools/service_breakdown.py > $X/service_breakdown_noflush.jsonl 2> $X/service_breakdown_noflush.err
python tools/fuzz_gpu.py 100 11 > $X/fuzz_product.json 2>&1
SLICESLICE_HIP_LIB=$L/libsliceslice_hip_tuning.so python tools/fuzz_gpu.py 100 12 > $X/fuzz_tuning.json 2>&1
python tools/fuzz_batched.py 60 13 > $X/fuzz_batched.json 2>&1
python tools/batch_probe.py --reps 20 1024x1048576 256x4194304 64x16777216 16x67108864 16384x65536 4096x1048576 65536x65536 > $X/batch_plan_probe.jsonl 2>/dev/null
python tools/batch_probe.py --find --reps 20 1024x1048576 64x16777216 16384x65536 4096x1048576 >> $X/batch_plan_probe.jsonl 2>/dev/null
for k in 1 2 3; do SLICESLICE_HIP_LIB=$L/libsliceslice_hip_tuning.so python tools/occ_probe.py --gib 1,8 > $X/occ_probe_run$k.jsonl 2>/dev/null; done
SLICESLICE_HIP_LIB=$L/libsliceslice_hip_tuning.so python tools/mode3_probe.py --gib 1,8 > $X/mode3_probe.jsonl 2>/dev/null
SLICESLICE_HIP_LIB=$L/libsliceslice_hip_tuning.so python tools/triple_probe.py > $X/triple_probe.jsonl 2>/dev/null
SLICESLICE_HIP_LIB=$L/libsliceslice_hip_tuning.so python tools/batch_triple_probe.py > $X/batch_triple_probe.jsonl 2>/dev/null
SLICESLICE_RCCL_LIB=$R/tests/native/libfake_rccl.so tools/native_bench set 8 8 200 2>/dev/null | grep '^{' > $X/native_set8.json
SLICESLICE_HIP_LIB=$L/libsliceslice_hip_tuning.so python tools/survival_probe.py > $X/survival_probe.jsonl 2>/dev/null
SLICESLICE_AUTOTUNE=0 python $R/bench.py --haystack-gib 8 --no-cpu-baseline --no-traffic > $X/bench8g_autotune_off.json 2> $X/bench8g_autotune_off.err
tail -3 $X/capture.log
