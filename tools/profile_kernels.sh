#!/bin/bash
# Per-kernel rocprofv3 evidence for the NON-headline kernels (run through gpurun from the repo root):
#   pass 1  --kernel-trace --stats  (x3)    -> duration per launch (median / min / mean of the measured launches), three processes
#   pass 2  --pmc FETCH_SIZE                -> HBM bytes fetched per launch (x2 gfx950 correction, KiB units)
#   pass 3  --pmc SQ_* (instruction mix)    -> VALU / SALU / LDS / VMEM per KiB piece, wait fraction
#   pass 4  NO profiler                     -> the hipEvent median of the same launches as a caller sees them (under rocprofv3 the
#                                              events read 5-8 % longer: the second clock of kernels.md)
# Counters are collected in their own runs, with --kernel-trace only (no other trace domains).
# Outputs under gpurun_out/<tag>_<case>_{kt,pmc,sq}/ ; tools/collect_kernel_profiles.py summarises them.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-kp}
shift || true
CASES=${@:-headline1g onebyte mode2 far_pair long_new long_wp find batched batched_plan text_worst text_wp text_refpair text_spaces text_spaces_new text_common_new random_text_needle text_refpair_static text_spaces_static text_worst_static}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in $CASES; do
  # THREE processes per case (one process per case cannot tell a 2 % regression from placement noise): the
  # table reports the median of the three processes' medians and their spread
  for rep in "" b c; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${c}_kt$rep -o r -- python $R/tools/profile_case.py $c 24 > $OUT/${TAG}_${c}_kt$rep.log 2>&1
  done
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_${c}_pmc -o r -- python $R/tools/profile_case.py $c 8 > $OUT/${TAG}_${c}_pmc.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_${c}_sq -o r -- python $R/tools/profile_case.py $c 8 > $OUT/${TAG}_${c}_sq.log 2>&1
  python $R/tools/profile_case.py $c 24 > $OUT/${TAG}_${c}_noprof.log 2>&1
  grep -h '^{"case"' $OUT/${TAG}_${c}_kt.log | tail -1
done
