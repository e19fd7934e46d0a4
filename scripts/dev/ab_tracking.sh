# usage: bash scripts/dev/ab_tracking.sh libA.so libB.so "C2 C4" [reps] -- alternating plain bench runs: ms per mapping step and per tracking iteration
A=$1; B=$2; CFGS=${3:-"C2 C4"}; N=${4:-3}
for c in $CFGS; do for i in $(seq $N); do for L in $A $B; do
FSGS_LIB_PATH=$L python bench.py --config $c --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$c', '$L'.split('/')[-1], 'ms/step %.4f tracking %.4f' % (d['ms_per_step'], d['tracking_step']['ms_per_iter']))"
done; done; done
