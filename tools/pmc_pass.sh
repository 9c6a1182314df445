#!/bin/bash
# usage (on the GPU box): tools/pmc_pass.sh TAG "COUNTER ..." [kernel-filter]
# One rocprofv3 --pmc pass (counters of one pass only, no other trace domain) over a short bench run;
# per-kernel averages -> gpurun_out/pmc_TAG.csv
TAG=$1; CNT=$2; FILT=${3:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
D=/tmp/pmc_$TAG
rm -rf $D
rocprofv3 --pmc $CNT --kernel-trace -d $D -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$TAG.log 2>&1
DB=$(find $D -name "*.db" | head -1)
python3 tools/rocpd_pmc.py $DB gpurun_out/pmc_$TAG.csv > /dev/null 2>&1
echo "== $TAG"; head -1 gpurun_out/pmc_$TAG.csv; grep -E "$FILT" gpurun_out/pmc_$TAG.csv | head -12
