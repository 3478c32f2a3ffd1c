"""Can RCCL run two ranks on ONE GPU on this box?  (It would let the library's communicator and the
sharded step run with world_size 2 over RCCL without a second GPU.)  Launch:
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 tools/rccl_same_gpu_probe.py"""
import os
import sys

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=int(os.environ["WORLD_SIZE"]),
                            device_id=torch.device("cuda", 0))
    t = torch.full((1024,), float(rank + 1), device="cuda:0")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce over RCCL on one shared GPU -> {float(t[0])}", flush=True)
    dist.destroy_process_group()
except Exception as e:        # expected: "Duplicate GPU detected"
    print(f"rank {rank}: RCCL refused: {str(e)[:300]}", flush=True)
    sys.exit(0)
