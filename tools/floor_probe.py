"""Back-to-back launch floor on this box: a trivial kernel at the gather+score kernel's grid (2016 x 256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for n in (256, 2016 * 256, 4 * 2016 * 256):
    x = torch.zeros(n, device='cuda')
    for _ in range(20):
        lib.ps_zero_floats(x.data_ptr(), n, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(2000):
        lib.ps_zero_floats(x.data_ptr(), n, st)
    e1.record(); torch.cuda.synchronize()
    print("zero %8d floats: %.2f us per launch" % (n, e0.elapsed_time(e1) / 2000 * 1e3))
