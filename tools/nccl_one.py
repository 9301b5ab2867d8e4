"""Rehearsal of the N>1 bench path's RCCL calls on the one-GPU box: a world-1 'nccl' group, the flat gradient buffer through
all_reduce, the fixed-capacity row messages through all_gather_into_tensor, barrier, and the MAX reduction of the timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
import bench
from prodsearch_amd import dist as pdist
sys.argv = ['bench.py', '--steps', '5', '--warmup', '2']
a = bench.parse()
wl = bench.TemWorkload(a, 0, torch.device('cuda', 0))
model, optim = wl.model, wl.optim
pdist.broadcast_parameters(model)
model.train()
for i in range(3):
    loss = wl.forward(i); model.zero_grad(); loss.backward()
    before = model._grad_flat.clone()
    dist.all_reduce(model._grad_flat, op=dist.ReduceOp.SUM)
    assert torch.equal(before, model._grad_flat)
    optim.step()
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t[0]) == 1.5
x = torch.arange(1000, device='cuda', dtype=torch.int64); out = torch.empty(1, 1000, device='cuda', dtype=torch.int64)
pdist._all_gather_flat(out, x, 1, None); assert torch.equal(out[0], x)
v = torch.randn(1000, 128, device='cuda'); outv = torch.empty(1, 1000, 128, device='cuda')
pdist._all_gather_flat(outv, v, 1, None); assert torch.equal(outv[0], v)
print("nccl world-1 rehearsal ok: loss %.5f, grad buffer %d floats" % (float(loss), model._grad_flat.numel()))
dist.destroy_process_group()
