#!/bin/bash
# Development aid (runs on the GPU box): HBM traffic per launch of a few bench cases -- FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3
# passes (together they exceed what one pass can collect), --kernel-trace only.   usage: tools/pmc_quick.sh <tag> <case> ...
set -u
export TMPDIR=/tmp
TAG=$1; shift
for CASE in "$@"; do
  OUT=gpurun_out/$TAG/pmc_$CASE; mkdir -p $OUT
  ARGS="--case $CASE --steps 60 --warmup 30 --no-cpu-baseline --no-secondary"
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/$CTR -- python3 bench.py $ARGS > $OUT/$CTR.log 2>&1 || echo "rocprofv3 $CTR $CASE failed"
  done
  python3 - <<PY
import csv, glob, statistics
c={}
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "step_" in r["Kernel_Name"]: c.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
fs=statistics.mean(c.get("FETCH_SIZE",[0]))*1024; ws=statistics.mean(c.get("WRITE_SIZE",[0]))*1024
algo=(939 if "$CASE".startswith("reach") else 1363)*8192
print("$CASE", "read(x2) %.1f MB write %.1f MB total %.1f MB = %.2fx algorithmic" % (2*fs/1e6, ws/1e6, (2*fs+ws)/1e6, (2*fs+ws)/algo), "launches", len(c.get("FETCH_SIZE",[])))
PY
done
