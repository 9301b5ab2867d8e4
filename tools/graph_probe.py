"""Do HIP graphs shorten the gap between dependent tiny kernels on this stack?  n dependent launches, stream vs graph."""
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
