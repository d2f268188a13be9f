cd $GRAFT_REPO_ROOT
(time timeout 1800 python -m pytest tests -m gpu -x -q) > gpurun_out/r2f_all.log 2>&1
tail -n 6 gpurun_out/r2f_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
timeout 600 python bench.py > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err; echo "bench rc=$?"
cut -c1-330 gpurun_out/r2f_bench.json
