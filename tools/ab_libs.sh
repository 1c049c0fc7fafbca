#!/bin/bash
# usage: tools/ab_libs.sh "<lib-a> <lib-b> ..." "<case steps warmup>" ...   -- alternate builds inside one GPU call (bench.py --case)
libs=$1; shift
cases=("$@")
for rep in 1 2; do
for c in "${cases[@]}"; do set -- $c
  for lib in $libs; do
    MCG_LIB=$lib timeout -k 10 150 python bench.py --case $1 --steps $2 --warmup $3 2>gpurun_out/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1 $lib ms/step %.4f' % d['ms_per_step'])"
  done
done
done
