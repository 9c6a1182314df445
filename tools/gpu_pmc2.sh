#!/bin/bash
# usage (on the GPU box): tools/gpu_pmc2.sh TAG "CTR1 CTR2 ..." ["CTRA CTRB" ...]
# one rocprofv3 --pmc pass per argument (counters of one argument share a pass) over a short bench
# run; per-kernel averages -> gpurun_out/${R}_pmc2_TAG_<n>.csv
R=${R:-r03}
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
n=0
for C in "$@"; do
  D=/tmp/pmc2_${TAG}_$n
  rm -rf $D
  rocprofv3 --pmc $C --kernel-trace -d $D -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2_${TAG}_$n.log 2>&1
  DB=$(find $D -name "*.db" | head -1)
  python3 tools/rocpd_pmc.py $DB gpurun_out/${R}_pmc2_${TAG}_$n.csv > /dev/null 2>&1
  echo "== $C"; head -9 gpurun_out/${R}_pmc2_${TAG}_$n.csv
  n=$((n+1))
done
