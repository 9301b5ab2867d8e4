"""n dependent tiny torch kernels, stream launches vs one HIP-graph replay.  NOTE: the stream arm is HOST-bound (torch's
~4 us per op), so the gap it shows (4.7 vs 1.9 us per kernel) is launch cost on the host, not a device-side gap: replaying
this library's step as graphs cut host time per step but not the device timeline (csrc/graph.h)."""
import torch
x = torch.zeros(4096, device='cuda')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n in (2, 4, 8, 16, 30):
    def chain():
        for _ in range(n):
            x.add_(1.0)
    for _ in range(5): chain()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50): chain()
    e1.record(); torch.cuda.synchronize()
    t_s = e0.elapsed_time(e1) * 1e3 / 50
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            chain()
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50): g.replay()
    e1.record(); torch.cuda.synchronize()
    t_g = e0.elapsed_time(e1) * 1e3 / 50
    print("n=%2d  stream %.1f us (%.2f/kernel)   graph %.1f us (%.2f/kernel)" % (n, t_s, t_s / n, t_g, t_g / n))
