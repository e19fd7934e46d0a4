"""CPU cost of one StagedLane.prefetch (+ the lookup that follows) and the duration of the copy it enqueues"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "free-surgs_amd"))
import torch
from fsgs_amd.staging import StagedLane

n, shape = 64, (3, 1024, 1280)
host = [torch.rand(shape).pin_memory() for _ in range(n)]
lane = StagedLane(host, "cuda", capacity=4, copy_stream=torch.cuda.Stream())
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(n):
        lane.prefetch(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for i in range(n - 4, n):
        lane[i]
    t3 = time.perf_counter()
    print("prefetch: %.1f us of CPU each; %d copies of %.1f MB done after %.2f ms (%.1f GB/s); resident lookup %.1f us" % (
        (t1 - t0) / n * 1e6, n, host[0].numel() * 4 / 1e6, (t2 - t0) * 1e3, n * host[0].numel() * 4 / (t2 - t0) / 1e9, (t3 - t2) / 4 * 1e6))
