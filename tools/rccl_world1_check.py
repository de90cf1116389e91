import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
x=torch.arange(12,dtype=torch.float32,device="cuda").view(3,4)
out=torch.empty((3,4),device="cuda")
dist.all_gather_into_tensor(out,x)
s=torch.ones(4,device="cuda"); dist.all_reduce(s,op=dist.ReduceOp.SUM); m=torch.tensor([5],device="cuda"); dist.all_reduce(m,op=dist.ReduceOp.MIN)
v=torch.ones(3,dtype=torch.float64,device="cuda"); dist.all_reduce(v,op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
print("rccl ok", bool(torch.equal(out,x)), s.tolist(), m.item(), v.tolist())
dist.destroy_process_group()
# the framebuffer gather's pack / collective / unpack on device tensors over RCCL (one rank: it must give back what it was given)
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MASTER_PORT"] = "29534"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from raytracing_amd import tiles
H, W = 1080, 1920
acc = torch.rand((H * W, 4), device="cuda"); aux = torch.rand((H * W, 4), device="cuda")
a0, x0 = acc.clone(), aux.clone()
tiles._gather_interleaved([acc, aux], H, W, tiles.TILE)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10):
    tiles._gather_interleaved([acc, aux], H, W, tiles.TILE)
torch.cuda.synchronize()
print("gather ok", bool(torch.equal(acc, a0) and torch.equal(aux, x0)), f"{(time.perf_counter() - t0) * 100:.2f} ms per gather of both 1080p framebuffers (one rank: pack + collective + unpack)")
dist.destroy_process_group()
