#!/bin/bash
# The evidence set of a round, one GPU-box visit:  gpurun -- 'bash scripts/gpu_profile_round.sh'
# bench lines (C2 default incl. cpu_baseline, C1, C4, C4 with densification, per-kernel events), rocprofv3 kernel stats,
# PMC traffic (separate passes), SQ counters, step / tracking timelines.  Everything lands in gpurun_out/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py 2>gpurun_out/bench_stderr.log | tail -1 > gpurun_out/bench.json
python bench.py --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_profile_all.json
python bench.py --config C1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_C1.json
python bench.py --config C4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_C4.json
python bench.py --config C4 --densify-every 300 --steps 600 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_C4_densify.json
python bench.py --dp-path --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 > gpurun_out/bench_dp_path.json
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-tracking --no-extras > gpurun_out/rocprof.log 2>&1
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) gpurun_out/bench_kernel_stats.csv
# the same evidence for C1 (BASELINE configs[0]: the four-waves-per-tile backward) and for the dense scene
for v in "C1:--config C1" "dense:--scene dense"; do
  n=${v%%:*}; a=${v#*:}
  rm -rf /tmp/prof_$n && mkdir -p /tmp/prof_$n
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o bench -- python bench.py $a --steps 50 --warmup 5 --no-cpu-baseline --no-tracking --no-extras > gpurun_out/rocprof_$n.log 2>&1
  cp $(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1) gpurun_out/bench_kernel_stats_$n.csv
done
python bench.py --scene dense --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_dense_profile_all.json
TRACE_ARGS="--config C1" bash scripts/gpu_trace.sh > /dev/null 2>&1; cp gpurun_out/trace_step.txt gpurun_out/trace_step_C1.txt
bash scripts/gpu_pmc.sh > gpurun_out/pmc.log 2>&1
bash scripts/gpu_sq.sh > gpurun_out/sq.log2 2>&1
bash scripts/gpu_trace.sh > /dev/null 2>&1
bash scripts/gpu_trace_tracking.sh > /dev/null 2>&1
bash scripts/gpu_trace_two_view.sh > /dev/null 2>&1
ls -la gpurun_out | tail -30
