#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
export IRA_FFN12=${IRA_FFN12:-0}
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/pmc_ffn -o a -- python3 $R/tools/time_ffn.py > $OUT/pmc_ffn_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_FLAT SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/pmc_ffn -o b -- python3 $R/tools/time_ffn.py > $OUT/pmc_ffn_b.log 2>&1
ls $OUT/pmc_ffn
