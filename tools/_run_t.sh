cd $GRAFT_REPO_ROOT
tools/native_bench sharded 8 100 > gpurun_out/r2i_sharded8g.json 2> gpurun_out/r2i_sharded.err
tools/native_bench sharded 0.001 2000 > gpurun_out/r2i_sharded1m.json 2>> gpurun_out/r2i_sharded.err
cat gpurun_out/r2i_sharded8g.json gpurun_out/r2i_sharded1m.json; tail -n 3 gpurun_out/r2i_sharded.err
