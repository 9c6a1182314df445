#!/bin/bash
# usage (GPU box): tools/pmc_cfg.sh c4|c5 [R]  -> profiles/${R}_bench_<cfg>_pmc_{fetch_size,write_size,mfma_busy}.csv
# three separate rocprofv3 --pmc passes (counters of one pass only) of a short bench run of another BASELINE config
CFG=$1; R=${2:-r06}
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
for pass in "fetch_size FETCH_SIZE" "write_size WRITE_SIZE" "mfma_busy SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  D=/tmp/pmc_${CFG}_$name; rm -rf $D
  rocprofv3 --pmc $* --kernel-trace -d $D -o r -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${CFG}_$name.log 2>&1
  DB=$(find $D -name "*.db" | head -1)
  python3 tools/rocpd_pmc.py $DB profiles/${R}_bench_${CFG}_pmc_$name.csv > /dev/null 2>&1
  head -8 profiles/${R}_bench_${CFG}_pmc_$name.csv | cut -c1-150
done
cp profiles/${R}_bench_${CFG}_pmc_*.csv gpurun_out/
