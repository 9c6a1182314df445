export TMPDIR=/tmp
D=/tmp/kstat_tl; rm -rf $D
rocprofv3 --kernel-trace -d $D -o r -- python3 tools/train_loop_bench.py > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log
python3 tools/rocpd_stats.py $(find $D -name "*.db" | head -1) | grep -E "at::|elementwise|reduce|copy|fill|index|gather|scatter" | head -30
