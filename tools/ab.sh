#!/bin/bash
# usage: tools/ab.sh "ENV1=.. ENV2=.." "ENVB=.." ...   one bench run per argument (per-class ms)
for v in "$@"; do
  echo "== $v"; env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], {k.split('<')[0][:14]+('L' if 'LSTM' in k else ''):(v['ms'],v['launches']) for k,v in j['roofline']['classes'].items()})"
done
