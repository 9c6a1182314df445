#!/bin/bash
# usage: [CFG="--batch 32"] [STEPS=40] tools/ab_ms.sh "ENV..." "ENV..."  -> ms per iteration only, interleaved twice ("base" = no override)
STEPS=${STEPS:-40}
for rep in 1 2; do
for v in "$@"; do
  e=$v; [ "$v" = base ] && e="MARL_NOP=1"
  ms=$(env $e python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline $CFG 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$CFG | ${v:0:40} | $ms"
done
done
