for v in 256 150; do
  echo "== MARL_NT_MIN_BLOCKS128=$v"; MARL_NT_MIN_BLOCKS128=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], {k:v['ms'] for k,v in j['roofline']['classes'].items()})"
done
