#!/bin/bash
# Instruction mix per kernel of the native DQN step (two counter passes, each its own rocprofv3 run).
#   bash tools/probes/inst_mix.sh fp32|bf16   -> gpurun_out/instmix_<precision>.txt
set -e
P=${1:-fp32}
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/instmix_$P.txt
: > $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY \
    -d /tmp/im_a -o p -- python3 $R/tools/pmc_step.py --batch 128 --steps 10 --precision $P > /tmp/im_a.log 2>&1
python3 $R/tools/pmc_db.py $(find /tmp/im_a -name '*_results.db' | head -1) >> $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY \
    -d /tmp/im_b -o p -- python3 $R/tools/pmc_step.py --batch 128 --steps 10 --precision $P > /tmp/im_b.log 2>&1 || { tail -5 /tmp/im_b.log; exit 1; }
python3 $R/tools/pmc_db.py $(find /tmp/im_b -name '*_results.db' | head -1) >> $OUT
