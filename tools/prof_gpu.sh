#!/bin/bash
# Run on the GPU box: kernel trace + PMC passes of one bench step. Outputs under gpurun_out/prof_*.
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python3 $ARGS > $OUT/prof_trace.log 2>&1 || { echo trace failed; tail -5 $OUT/prof_trace.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- python3 $ARGS > $OUT/prof_fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/prof_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- python3 $ARGS > $OUT/prof_write.log 2>&1 || { echo write failed; tail -5 $OUT/prof_write.log; exit 1; }
find $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write -name "*.csv" | head -20
# keep only the summaries small enough to merge back (<64 MiB)
du -sh $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write
