#!/bin/bash
# Run on the GPU box: every bench.py workload once, one JSON line each into gpurun_out/bench_<prefix>_<name>.json (copied to profiles/ afterwards).
# usage: tools/bench_all.sh [prefix]
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out; mkdir -p $OUT
PFX=${1:-r02}
cd $R
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $OUT/bench_${PFX}_$name.json 2> $OUT/bench_${PFX}_$name.err || { echo "$name failed"; tail -3 $OUT/bench_${PFX}_$name.err; return 1; }
        python -c "import json,sys; d=json.loads(open('$OUT/bench_${PFX}_$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], 'ms', d['value'], d['unit'], 'psnr_delta_db', d.get('psnr_delta_db'))"; }
run restormer4k --steps 10 --warmup 3 && \
run restormer_5tiles --steps 10 --warmup 3 --width 2432 --height 512 --no-cpu-baseline && \
run restormer_10tiles --steps 10 --warmup 3 --width 2432 --height 992 --no-cpu-baseline && \
run swinir_x4 --workload swinir_x4 --steps 5 --warmup 2 && \
run nafnet512 --workload nafnet512 --steps 20 --warmup 5 && \
run restormer4k_whole --workload restormer4k_whole --steps 5 --warmup 2 --no-cpu-baseline && \
run xrestormer720p --workload xrestormer720p --steps 5 --warmup 2 --no-cpu-baseline && \
run mprnet720p --workload mprnet720p --steps 5 --warmup 2 --no-cpu-baseline && \
run realesrgan_x4 --workload realesrgan_x4 --steps 5 --warmup 2 --no-cpu-baseline && \
run agent_chain --workload agent_chain --steps 3 --warmup 1 --no-cpu-baseline
