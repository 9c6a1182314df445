#!/bin/bash
# usage (on the GPU box): tools/gpu_pmc.sh TAG
# Three separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES) over a
# short bench run; per-kernel averages -> gpurun_out/${R}_pmc_<counter>_TAG.csv
R=${R:-r03}
TAG=$1
mkdir -p gpurun_out
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  D=/tmp/pmc_${TAG}_$C
  rm -rf $D
  rocprofv3 --pmc $C --kernel-trace -d $D -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${TAG}_$C.log 2>&1
  DB=$(find $D -name "*.db" | head -1)
  python3 tools/rocpd_pmc.py $DB gpurun_out/${R}_pmc_${C}_$TAG.csv > /dev/null 2>&1
  echo "== $C"; head -8 gpurun_out/${R}_pmc_${C}_$TAG.csv
done
