#!/bin/bash
# usage: [CFG="--config c4"] [STEPS=40] tools/ab.sh "ENV1=.. ENV2=.." "ENVB=.." ...   one bench run per argument
# (per-class ms); "base" = no override.  Runs interleaved twice (A B A B) so that drift shows.
STEPS=${STEPS:-40}
for rep in 1 2; do
for v in "$@"; do
  e=$v; [ "$v" = base ] && e="MARL_NOP=1"
  echo -n "== $v: "; env $e python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline $CFG 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], {k.split('<')[0][:10]+('L' if 'LSTM' in k else ''):(v['ms'],v['launches']) for k,v in j['roofline']['classes'].items()})"
done
done
