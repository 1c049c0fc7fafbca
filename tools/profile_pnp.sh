#!/bin/bash
# GPU box: PickAndPlace bench lines + rocprofv3 kernel stats for the step_pnp kernel.  Outputs under gpurun_out/<tag>/
set -u
TAG=${1:-r01_pnp}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --task pnp --steps 300 --warmup 55 > $OUT/bench_pnp_joint.json 2> $OUT/err1.log
python3 bench.py --task pnp --dr --steps 300 --warmup 55 > $OUT/bench_pnp_dr.json 2> $OUT/err2.log
python3 bench.py --task pnp --controller IK --steps 100 --warmup 20 > $OUT/bench_pnp_ik.json 2> $OUT/err3.log
python3 bench.py --task pnp --scripted-grasp --steps 60 --warmup 5 > $OUT/bench_pnp_grasp.json 2> $OUT/err4.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --task pnp --steps 100 --warmup 55 > $OUT/stats.log 2>&1
PARGS="--task pnp --steps 100 --warmup 55 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $PARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py $PARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py $PARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d $OUT/pmc_flops -- python3 bench.py $PARGS > $OUT/pmc_flops.log 2>&1
cat $OUT/bench_pnp_joint.json
