#!/bin/bash
# GPU box: effective clock (GRBM_GUI_ACTIVE) and MFMA-busy at that clock of the ablation builds.
# needs tools/bin/libmarl_abl.so (tools/build_variant.sh abl -DMARL_G3_ABLATE)
set -uo pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
for t in "$@"; do
  D=/tmp/clk_$t; rm -rf $D
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $D -o r -- python3 tools/$t.py > gpurun_out/clk_$t.log 2>&1
  DB=$(find $D -name "*.db" | head -1)
  [ -n "$DB" ] || { echo "no rocpd db for $t"; tail -5 gpurun_out/clk_$t.log; continue; }
  python3 tools/rocpd_clock.py $DB gpurun_out/r06_clock_$t.csv 15 | cut -c1-170
done
