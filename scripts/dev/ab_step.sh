# usage: bash scripts/dev/ab_step.sh libA.so libB.so [reps] [bench args...] -- alternating bench runs, ms/step and blend kernel us of each
A=$1; B=$2; N=${3:-3}; shift 3 || true
for i in $(seq $N); do
  for L in "$A" "$B"; do
    FSGS_LIB_PATH=$L python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-tracking "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$L'.split('/')[-1], 'ms/step %.4f' % d['ms_per_step'], ' '.join('%s=%.1f' % (n, 1e3*v['avg_ms']) for n, v in k.items()))"
  done
done
