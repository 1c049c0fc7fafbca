#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes (HBM traffic, SQ, FP64 flops) for one
# bench.py workload.   Usage: tools/profile_case.sh <tag> <case> [steps] [warmup]     (outputs under gpurun_out/<tag>/<case>/)
# Counters are collected in their own passes with --kernel-trace only (never with --sys-trace etc.).
set -u
TAG=${1:-r02}
CASE=${2:-reach-joint}
STEPS=${3:-200}
WARM=${4:-60}
OUT=gpurun_out/$TAG/$CASE
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--case $CASE --steps $STEPS --warmup $WARM --no-cpu-baseline --no-secondary"
python3 -c "from mycobotgym_amd.build import source_hash; print(source_hash())" > $OUT/src_sha256.txt
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d $OUT/pmc_flops -- python3 bench.py $ARGS > $OUT/pmc_flops.log 2>&1
cat $OUT/bench.json
