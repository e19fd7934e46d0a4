# usage: bash scripts/dev/ab_bench.sh [bench args...]   -- one profile-all line, compact
python bench.py --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('ms/step %.4f'%d['ms_per_step'], ' '.join('%s=%.1f'%(n,1e3*v['avg_ms']) for n,v in sorted(k.items(), key=lambda kv:-kv[1]['avg_ms'])), 'tracking %.4f' % (d.get('tracking_step') or {}).get('ms_per_iter', 0))"
