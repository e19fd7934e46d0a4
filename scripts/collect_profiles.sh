#!/bin/bash
# After `gpurun -- 'bash scripts/gpu_profile_round.sh'`: copy the evidence set from gpurun_out/ into profiles/<tag>_* and
# rebuild profiles/<tag>_pmc_traffic.json (bench.py reads the newest one).   bash scripts/collect_profiles.sh r02
set -eu
cd "$(dirname "$0")/.."
tag=${1:-r02}
for n in bench bench_C1 bench_C4 bench_C4_densify bench_dp_path bench_profile_all; do cp gpurun_out/$n.json profiles/${tag}_$n.json; done
cp gpurun_out/bench_kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv
for n in C1 dense; do [ -f gpurun_out/bench_kernel_stats_$n.csv ] && cp gpurun_out/bench_kernel_stats_$n.csv profiles/${tag}_bench_kernel_stats_$n.csv; done
[ -f gpurun_out/bench_dense_profile_all.json ] && cp gpurun_out/bench_dense_profile_all.json profiles/${tag}_bench_dense_profile_all.json
[ -f gpurun_out/trace_step_C1.txt ] && cp gpurun_out/trace_step_C1.txt profiles/${tag}_step_timeline_C1.txt
cp gpurun_out/pmc_FETCH_SIZE_summary.csv profiles/${tag}_pmc_FETCH_SIZE_summary.csv
cp gpurun_out/pmc_WRITE_SIZE_summary.csv profiles/${tag}_pmc_WRITE_SIZE_summary.csv
cp gpurun_out/sq_summary.txt profiles/${tag}_sq_counters.txt
cp gpurun_out/trace_step.txt profiles/${tag}_step_timeline.txt
cp gpurun_out/trace_tracking.txt profiles/${tag}_tracking_timeline.txt
[ -f gpurun_out/trace_two_view.txt ] && cp gpurun_out/trace_two_view.txt profiles/${tag}_two_view_timeline.txt
for n in full_size_parity outlier_statistics; do [ -f gpurun_out/${tag}_$n.jsonl ] && cp gpurun_out/${tag}_$n.jsonl profiles/${tag}_$n.jsonl; done
R=$(python -c "import json; print(json.load(open('gpurun_out/bench.json'))['config']['num_rendered'])")
python scripts/make_pmc_json.py $tag $R 300000 1280 1024 | grep blend
grep -o '"avg_kernel_ms": [0-9.]*' gpurun_out/rocprof.log | head -1
grep "blend_bwd_kernel<6, true" profiles/${tag}_bench_kernel_stats.csv | awk -F, '{print "rocprofv3 blend_bwd avg ns:", $(NF-4)}'
