for i in 1 2 3; do for L in ${LIBS:-libfsgs_hip.so diag/libfsgs_hip.u8.so diag/libfsgs_hip.u2.so}; do
FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-tracking --profile-all 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$L', 'ms/step %.4f' % d['ms_per_step'], 'scatter %.1f sort %.1f' % (1e3*k['sort_depth']['avg_ms'], 1e3*k['sort_tile']['avg_ms']))"
done; done
