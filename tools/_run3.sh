set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 1800 python -m pytest tests -m gpu -q -x) > gpurun_out/r2c_alltests.log 2>&1
echo "alltests rc=$?" >> gpurun_out/r2c_alltests.log
GD=tests/golden/data
for sp in 1 0; do
  SLICESLICE_SPIN_WAIT=$sp tools/native_bench latency 2000 > gpurun_out/r2c_latency_spin$sp.json 2> gpurun_out/r2c_latency_spin$sp.err
  SLICESLICE_SPIN_WAIT=$sp tools/native_bench config1 $GD/i386.txt $GD/words.txt 5 > gpurun_out/r2c_config1_spin$sp.json 2>> gpurun_out/r2c_latency_spin$sp.err
done
tail -n 4 gpurun_out/r2c_alltests.log
cat gpurun_out/r2c_latency_spin1.json gpurun_out/r2c_latency_spin0.json gpurun_out/r2c_config1_spin1.json gpurun_out/r2c_config1_spin0.json
