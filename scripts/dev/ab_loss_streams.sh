#!/bin/bash
# A/B of the loss stage's layout on one box: FSGS_LOSS_STREAMS=2 (photometric pair || Pearson chain on two streams, the default)
# against FSGS_LOSS_STREAMS=1 (round 6: two launches on one stream, fsgs_view_losses_forward_backward), alternating, + the timeline of one step of each.
#   gpurun -- 'bash scripts/dev/ab_loss_streams.sh [tag]'
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
tag=${1:-ab_loss}
python -m pytest tests/test_loss_gpu.py tests/test_fast_step_gpu.py tests/test_harness_pin_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -15 > gpurun_out/${tag}_pytest.log
tail -3 gpurun_out/${tag}_pytest.log
out=gpurun_out/${tag}.txt; : > $out
for rep in 1 2 3; do
  for mode in 2 1; do
    for cfg in C2 C1; do
      FSGS_LOSS_STREAMS=$mode python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-harness 2>/dev/null | tail -1 > /tmp/line.json
      python - $mode $cfg $rep <<'PY' | tee -a $out
import json, sys
d = json.load(open("/tmp/line.json"))
tr = d.get("tracking_step") or {}
print("streams=%s %s rep %s: ms/step %.4f  blocks %s  tracking %.4f" % (sys.argv[1], sys.argv[2], sys.argv[3], d["ms_per_step"],
      "%.4f..%.4f" % (d["timed_blocks"]["ms_per_step_min"], d["timed_blocks"]["ms_per_step_max"]), tr.get("ms_per_iter", 0.0)))
PY
    done
  done
done
FSGS_LOSS_STREAMS=2 bash scripts/gpu_trace.sh > /dev/null 2>&1; cp gpurun_out/trace_step.txt gpurun_out/${tag}_timeline_two_streams.txt
FSGS_LOSS_STREAMS=1 bash scripts/gpu_trace.sh > /dev/null 2>&1; cp gpurun_out/trace_step.txt gpurun_out/${tag}_timeline_one_stream.txt
cat gpurun_out/${tag}_timeline_one_stream.txt
