#!/bin/bash
# SQ counters of one bench step (two rocprofv3 --pmc passes).  usage: tools/pmc_sq.sh [prefix] [bench args...]
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out; mkdir -p $OUT
PFX=${1:-prof}; shift
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile $*"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/${PFX}_sq -o sq -- python3 $ARGS > $OUT/${PFX}_sq.log 2>&1 || { echo failed; tail -5 $OUT/${PFX}_sq.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${PFX}_sq2 -o sq2 -- python3 $ARGS > $OUT/${PFX}_sq2.log 2>&1 || { echo failed2; tail -5 $OUT/${PFX}_sq2.log; }
ls $OUT/${PFX}_sq $OUT/${PFX}_sq2
