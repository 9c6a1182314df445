#!/bin/bash
# usage: tools/ab_so.sh BASE.so [rounds]   A/B of two builds inside one GPU call: alternates the
# committed-tree library (BASE.so) with the working-tree one, prints ms per iteration of each run
BASE=$1; N=${2:-3}
cp marlclassification_amd/csrc/libmarl_hip.so /tmp/new.so
for i in $(seq $N); do
  for v in base new; do
    if [ $v = base ]; then cp $BASE marlclassification_amd/csrc/libmarl_hip.so; else cp /tmp/new.so marlclassification_amd/csrc/libmarl_hip.so; fi
    echo -n "$v "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
cp /tmp/new.so marlclassification_amd/csrc/libmarl_hip.so
