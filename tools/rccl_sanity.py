"""One-rank RCCL sanity check on a single-GPU box: communicator init, all_reduce of the flat gradient buffer's size,
barrier and the float64 MAX all-reduce bench.py uses — the calls of the N>1 path, on backend 'nccl' (= RCCL).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_sanity.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29511')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
x = torch.ones(6_760_000, device='cuda')
dist.all_reduce(x)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device='cuda')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    dist.all_reduce(x)
torch.cuda.synchronize()
print("rccl ok: world %d, all_reduce(27 MB) %.1f us/call (host+device), max %.1f"
      % (dist.get_world_size(), (time.perf_counter() - t0) * 1e4, float(t[0])))
dist.destroy_process_group()
