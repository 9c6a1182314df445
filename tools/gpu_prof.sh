#!/bin/bash
# usage (on the GPU box): tools/gpu_prof.sh TAG [bench.py args...]
# rocprofv3 kernel trace of a short bench run -> gpurun_out/${R}_stats_TAG.csv (+ launch sequence)
R=${R:-r03}
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
D=/tmp/prof_$TAG
rm -rf $D
rocprofv3 --kernel-trace -d $D -o r -- python3 bench.py --steps 7 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/prof_$TAG.log 2>&1
DB=$(find $D -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB gpurun_out/${R}_stats_$TAG.csv > /dev/null 2>&1
python3 tools/rocpd_stats.py $DB /dev/null --seq "" 420 > gpurun_out/${R}_seq_$TAG.txt 2>&1
grep '"metric"' gpurun_out/prof_$TAG.log | cut -c1-400
head -24 gpurun_out/${R}_stats_$TAG.csv
