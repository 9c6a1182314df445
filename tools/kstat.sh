#!/bin/bash
# usage (on the GPU box): tools/kstat.sh PATTERN bench-args...   -> per-kernel averages matching PATTERN
PAT=$1; shift
export TMPDIR=/tmp
D=/tmp/kstat_$$; rm -rf $D
rocprofv3 --kernel-trace -d $D -o r -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > /tmp/kstat_$$.log 2>&1
grep '"metric"' /tmp/kstat_$$.log | cut -c1-160
python3 tools/rocpd_stats.py $(find $D -name "*.db" | head -1) | grep -E "$PAT"
