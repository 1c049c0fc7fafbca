#!/bin/bash
# Runs on the GPU box: SQ counter passes that separate instruction-fetch, memory-wait and issue time of the step kernel.
# Usage: tools/profile_stalls.sh <tag> [bench args]
set -u
TAG=${1:-stalls}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 100 --warmup 20 --no-cpu-baseline --no-secondary $*"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/p2 -- python3 bench.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/p3 -- python3 bench.py $ARGS > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p4 -- python3 bench.py $ARGS > $OUT/p4.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p5 -- python3 bench.py $ARGS > $OUT/p5.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"step_\w+<\d>", r["Kernel_Name"])
        if not m: continue
        k = m.group(0)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:24s} {v / n[(k, c)]:16.1f} per launch")
PY
