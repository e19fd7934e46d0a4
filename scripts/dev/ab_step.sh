# usage: bash scripts/dev/ab_step.sh libA.so libB.so [reps] -- alternating plain bench runs (no per-kernel events), ms/step of each
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in "$A" "$B"; do
    FSGS_LIB_PATH=$L python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], 'ms/step %.4f' % d['ms_per_step'])"
  done
done
