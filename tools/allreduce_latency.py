"""Per-call GPU time of dist.all_reduce on the step's accumulator sizes (run under torchrun)."""
import os, time, torch, torch.distributed as dist
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
for n in (1_300_104, 650_000, 20_000, 64):
    t = torch.ones(n, device=dev)
    for _ in range(5):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    if dist.get_rank() == 0:
        print(f"world {dist.get_world_size()} n={n} floats: {1e6*dt:.1f} us per all_reduce (back to back)")
dist.destroy_process_group()
