"""What bench.py's `comm.rccl_info` will contain: a 1-rank RCCL group on the one test GPU (the only RCCL communicator a
one-GPU box can build), the init-time INFO log parsed by bench.rccl_info_lines()."""
import os, sys
sys.path[:0] = [".", "free-surgs_amd"]
import torch, torch.distributed as dist
import bench
bench.rccl_diagnostics_env()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
for l in bench.rccl_info_lines() or []:
    print(l)
print({k: v for k, v in os.environ.items() if k.startswith("NCCL_")})
dist.destroy_process_group()
