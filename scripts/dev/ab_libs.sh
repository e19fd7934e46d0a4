# usage: bash scripts/dev/ab_libs.sh "<lib> <lib> ..." "<configs>" [reps] -- alternating plain bench runs of whole libraries: ms per mapping step / tracking iteration
LIBS=$1; CFGS=${2:-C2}; N=${3:-2}
for c in $CFGS; do for i in $(seq $N); do for L in $LIBS; do
FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$c', '$L', 'ms/step %.4f tracking %.4f' % (d['ms_per_step'], d['tracking_step']['ms_per_iter']))"
done; done; done
