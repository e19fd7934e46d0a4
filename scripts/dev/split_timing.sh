for n in 1 0; do
  if [ $n = 1 ]; then export FSGS_NO_SPLIT_BACKWARD=1; else unset FSGS_NO_SPLIT_BACKWARD; fi
  python bench.py --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('nosplit', $n, 'ms/step %.4f'%d['ms_per_step'], 'blend_bwd %.1f'%(1e3*k['blend_bwd']['avg_ms']), 'blend_fwd %.1f'%(1e3*k['blend_fwd']['avg_ms']), 'tracking %.4f' % d['tracking_step']['ms_per_iter'])"
done
