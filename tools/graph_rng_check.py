"""Does torch's CUDA-graph capture advance the philox stream correctly on this ROCm build?
Prints draws of two normals and a gamma across replays, and the eager draws after."""
import torch
torch.manual_seed(0)
dev = "cuda"
s = torch.cuda.Stream()
one = torch.ones(6, device=dev)
with torch.cuda.stream(s):
    torch.randn(6, device=dev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    a = torch.randn(6, device=dev)
    b = torch.randn(6, device=dev)
    c = torch._standard_gamma(one)
    d = torch._standard_gamma(one)
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, "a", a[:3].tolist(), "b", b[:3].tolist(), "c", c[:3].tolist(), "d", d[:3].tolist())
print("eager", torch.randn(3, device=dev).tolist())
