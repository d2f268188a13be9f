set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$PWD/sliceslice-rs_amd/csrc
LIBS="cur=$C/libsliceslice_hip.so nopad=$C/libsliceslice_hip.so@41 prev=$C/libsliceslice_hip_prev.so"
rm -f gpurun_out/ab9*.jsonl
timeout 900 python tools/ab_inproc.py --libs $LIBS --gib 1 --cases n16,tworst,tspaces,tpriv,tcommon,tmid --rounds 8 --reps 6 >> gpurun_out/ab9_1g.jsonl 2>> gpurun_out/ab9.err
timeout 900 python tools/ab_inproc.py --libs $LIBS --gib 64 --cases n16 --rounds 6 --reps 6 >> gpurun_out/ab9_64g.jsonl 2>> gpurun_out/ab9.err
grep median gpurun_out/ab9_*.jsonl
