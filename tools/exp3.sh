python -m pytest tests/test_gpu_episode.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -8
for v in 1 0; do
  echo "== MARL_CNN_FWD2=$v"; MARL_CNN_FWD2=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], {k:v['ms'] for k,v in j['roofline']['classes'].items()})"
done
