#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + separate PMC passes for HBM traffic.
# Usage: tools/profile_r.sh <round-tag>   (outputs under gpurun_out/<tag>/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
ARGS="--steps 200 --warmup 20 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d $OUT/pmc_flops -- python3 bench.py $ARGS > $OUT/pmc_flops.log 2>&1
find $OUT -name "*.csv" | head -40
cat $OUT/bench.json
