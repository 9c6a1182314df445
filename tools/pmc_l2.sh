#!/bin/bash
# GPU box: memory-side counters of the C3 bench (round 6: what bounds the L2-resident operand streams of the LSTM /
# in-loop NT launches) -> profiles/${R}_bench_c3_pmc_l2.csv, _ta.csv
set -uo pipefail
R=${R:-r06}
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
pass() {
  local name=$1; shift
  local D=/tmp/pmcl2_$name; rm -rf $D
  # (a pass with a counter the profiler rejects aborts and then HANGS: every pass runs under its own timeout)
  timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace -d $D -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_l2_$name.log 2>&1
  local DB; DB=$(find $D -name "*.db" | head -1)
  [ -n "$DB" ] || { echo "no db for $name"; grep -i "error\|invalid\|abort" gpurun_out/pmc_l2_$name.log | head -3; return; }
  python3 tools/rocpd_pmc.py $DB profiles/${R}_bench_c3_pmc_$name.csv | head -9 | cut -c1-170
  cp profiles/${R}_bench_c3_pmc_$name.csv gpurun_out/
}
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum TCC_CYCLE_sum
pass l2req TCC_REQ_sum TCC_READ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum
pass ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
