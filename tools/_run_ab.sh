set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$PWD/sliceslice-rs_amd/csrc
LIBS="cur=$C/libsliceslice_hip.so two=$C/libsliceslice_hip_two.so"
rm -f gpurun_out/ab4_sizes.jsonl
for g in 0.25 1 2 4 8 16 32 64; do
echo "{\"gib\": $g}" >> gpurun_out/ab4_sizes.jsonl
timeout 900 python tools/ab_inproc.py --libs $LIBS --gib $g --cases n16,n1 --rounds 4 | grep median >> gpurun_out/ab4_sizes.jsonl 2>> gpurun_out/ab4.err
done
cat gpurun_out/ab4_sizes.jsonl
# redo the rocprof passes of the headline without the extra configs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=r02cap
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_pmc $OUT/${TAG}_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $OUT/${TAG}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $OUT/${TAG}_pmc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_sq -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $OUT/${TAG}_sq.log 2>&1
