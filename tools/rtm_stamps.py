"""Diagnostic: phase timeline of one workgroup of the review transformer's forward gather (rtm_embed4_kernel; s_memtime per wave).
    PS_RTM_STAMP=1 python tools/rtm_stamps.py        (on the GPU box)"""
import argparse, ctypes, os, sys
os.environ['PS_DIAG_LIB'] = '1'      # stamps exist in the diagnostic build only (python -m prodsearch_amd.build --diag)
os.environ['PS_RTM_STAMP'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from prodsearch_amd import _lib

a = argparse.Namespace(workload='c4', encoder='pvc', dropout=0.1, row_sparse=False)
wl = bench.RtmWorkload(a, 'c4', 0, torch.device('cuda', 0))
wl.model.train()
raw = ctypes.CDLL(_lib.lib_path())
with torch.no_grad():
    for i in range(6):
        wl.forward(i)
    buf = torch.zeros(64, dtype=torch.int64, device='cuda')
    raw.ps_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    wl.forward(7)
    torch.cuda.synchronize()
    raw.ps_debug_set_stamp_buffer(ctypes.c_void_p(0))
t = buf.cpu().view(4, 16)
names = ['start', 'review ids / segments decoded', 'word ids, token masks, lists, dropout words', 'gather done', 'end']
live = [w for w in range(4) if int(t[w, 0])]
t0 = min(int(t[w, 0]) for w in live) if live else 0
print('%-46s' % 'phase' + ''.join('   wave%d' % w for w in range(4)) + '   (s_memtime ticks since the first wave started)')
for i, n in enumerate(names):
    print('%-46s' % n + ''.join('%8d' % (int(t[w, i]) - t0 if int(t[w, i]) else -1) for w in range(4)))
