cd $GRAFT_REPO_ROOT
RT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
export SLICESLICE_HIP_LIB=$PWD/sliceslice-rs_amd/csrc/libsliceslice_hip_asan.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0:halt_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
LD_PRELOAD=$RT python -X faulthandler -c "
import torch; print('torch', torch.cuda.is_available(), flush=True)
import sliceslice_rs_amd as ss; print(ss.lib(), flush=True)
s = ss.DynamicHipSearcher.new(b'abc'); print(s.search_in(b'xxabcxx'), flush=True)
" > gpurun_out/r2h_probe.log 2>&1; echo "probe rc=$?" >> gpurun_out/r2h_probe.log
LD_PRELOAD=$RT timeout 1200 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_filter_and_configs.py -m gpu -x -q -k "not 64gib and not beyond_4gib" > gpurun_out/r2h_asan_py.log 2>&1; echo "rc=$?" >> gpurun_out/r2h_asan_py.log
tail -n 15 gpurun_out/r2h_probe.log; tail -n 15 gpurun_out/r2h_asan_py.log
