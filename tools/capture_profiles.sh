#!/bin/bash
# Round capture on the GPU box (run through gpurun from the repo root): the default bench line, the
# rocprofv3 kernel trace of the same command, and the two PMC passes (FETCH_SIZE; SQ instruction mix).
# Outputs under gpurun_out/; tools/collect_profiles.py turns them into profiles/<round>/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-cap}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_pmc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_sq -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_sq.log 2>&1
python $R/tools/bench_configs.py > $OUT/${TAG}_configs.jsonl 2> $OUT/${TAG}_configs.err
cat $OUT/${TAG}_bench.json
