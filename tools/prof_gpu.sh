#!/bin/bash
# Run on the GPU box: kernel trace + PMC passes (separate runs, as the HBM section of MI355X_MICROARCH.md prescribes) of one
# bench step.  usage: tools/prof_gpu.sh [prefix] [extra bench.py args...]   -> gpurun_out/<prefix>_{trace,fetch,write}/
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out
PFX=${1:-prof}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${PFX}_trace -o trace -- python3 $ARGS > $OUT/${PFX}_trace.log 2>&1 || { echo trace failed; tail -5 $OUT/${PFX}_trace.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${PFX}_fetch -o fetch -- python3 $ARGS > $OUT/${PFX}_fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/${PFX}_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${PFX}_write -o write -- python3 $ARGS > $OUT/${PFX}_write.log 2>&1 || { echo write failed; tail -5 $OUT/${PFX}_write.log; exit 1; }
du -sh $OUT/${PFX}_trace $OUT/${PFX}_fetch $OUT/${PFX}_write
