python -m pytest tests -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>&1 | tail -1 > gpurun_out/final_bench.json
FSGS_DIST_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --smoke --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/final_bench_n2.json
