#!/bin/bash
# SQ / LDS counters of the fused GDFN kernels at C = 96, 8 x 512 x 512 (tools/time_ffn2.py runs generation 1 and 2 back to back).
# usage (GPU box): tools/pmc_ffn2.sh   ->  gpurun_out/pmc_ffn2/{a,b}_counter_collection.csv ; summary: tools/pmc_ffn2_summary.py
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/pmc_ffn2 -o a -- python3 $R/tools/time_ffn2.py 8 96 > $OUT/pmc_ffn2_a.log 2>&1 || { echo pass a failed; tail -3 $OUT/pmc_ffn2_a.log; }
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_ffn2 -o b -- python3 $R/tools/time_ffn2.py 8 96 > $OUT/pmc_ffn2_b.log 2>&1 || { echo pass b failed; tail -3 $OUT/pmc_ffn2_b.log; }
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_IFETCH SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ffn2 -o c -- python3 $R/tools/time_ffn2.py 8 96 > $OUT/pmc_ffn2_c.log 2>&1 || { echo pass c failed; tail -3 $OUT/pmc_ffn2_c.log; }
python3 $R/tools/pmc_ffn2_summary.py $OUT/pmc_ffn2
