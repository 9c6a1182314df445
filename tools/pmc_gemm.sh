#!/bin/bash
# usage (GPU box): tools/pmc_gemm.sh TAG "script args" "CTR1 CTR2 ..." ["CTRA ..."]  - one rocprofv3 --pmc
# pass per counter group over `python3 tools/ts_run.py <args>`; per-kernel averages printed
TAG=$1; ARGS=$2; shift; shift
export TMPDIR=/tmp
n=0
for C in "$@"; do
  D=/tmp/pmcg_${TAG}_$n; rm -rf $D
  rocprofv3 --pmc $C --kernel-trace -d $D -o r -- python3 tools/ts_run.py $ARGS > /tmp/pmcg.log 2>&1
  DB=$(find $D -name "*.db" | head -1)
  echo "== $C"; python3 tools/rocpd_pmc.py $DB | grep -i "split\|kernel,"
  n=$((n+1))
done
