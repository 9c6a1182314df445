for rb in 8 4 2 16; do
  echo "== MARL_CNN_RB=$rb"; MARL_CNN_RB=$rb MARL_CNN_LDS_KB=100 python bench.py --steps 8 --warmup 3 --no-cpu-baseline | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], {k:v['ms'] for k,v in j['roofline']['classes'].items()})"
done
echo "== c4@32 eager / graph"; python bench.py --config c4 --steps 20 --warmup 5 | cut -c1-160; python bench.py --config c4 --steps 20 --warmup 5 --graph | cut -c1-160
echo "== c2@32 eager / graph"; python bench.py --config c2 --batch 32 --steps 50 --warmup 10 | cut -c1-160; python bench.py --config c2 --batch 32 --steps 50 --warmup 10 --graph | cut -c1-160
