"""Diagnostic: when does every workgroup of the review transformer's forward gather (rtm_embed4_kernel) start and end?
    python tools/rtm_wg_times.py        (on the GPU box; PS_RTM_STAMP=1 PS_RTM_DIAG=64 are set here)
s_memtime per workgroup (wave 0) + its XCC id; times are compared inside one XCC only."""
import argparse, ctypes, os, sys
os.environ['PS_DIAG_LIB'] = '1'      # stamps exist in the diagnostic build only (python -m prodsearch_amd.build --diag)
os.environ['PS_RTM_STAMP'] = '1'
os.environ['PS_RTM_DIAG'] = '64'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from prodsearch_amd import _lib

a = argparse.Namespace(workload='c4', encoder='pvc', dropout=0.1, row_sparse=False)
wl = bench.RtmWorkload(a, 'c4', 0, torch.device('cuda', 0))
wl.model.train()
raw = ctypes.CDLL(_lib.lib_path())
NWG = 8192
with torch.no_grad():
    for i in range(6):
        wl.forward(i)
    buf = torch.zeros(64 + 3 * NWG, dtype=torch.int64, device='cuda')
    raw.ps_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    wl.forward(7)
    torch.cuda.synchronize()
    raw.ps_debug_set_stamp_buffer(ctypes.c_void_p(0))
t = buf.cpu().numpy()[64:].reshape(NWG, 3)
if len(sys.argv) > 1:
    np.save(sys.argv[1], t)
live = (t[:, 0] != 0) & (t[:, 2] < 8) & (t[:, 1] >= t[:, 0])
live[:48] = False                      # (the first rows share the buffer with other kernels' phase slots)
idx = np.nonzero(live)[0]
st, en = t[live, 0], t[live, 1]
t0 = st.min()
dur = en - st
print("workgroups that stamped: %d; 1 tick = 10 ns" % live.sum())
print("first start 0, last start %d, last end %d ticks; life min / median / p90 / max %d / %d / %d / %d"
      % (st.max() - t0, en.max() - t0, dur.min(), int(np.median(dur)), int(np.percentile(dur, 90)), dur.max()))
T = en.max() - t0
pts = [int(T * k / 20) for k in range(21)]
print("alive at k/20 of the span: " + " ".join("%4d" % int(((st - t0 <= p) & (en - t0 > p)).sum()) for p in pts))
print("started by k/20 of the span: " + " ".join("%4d" % int((st - t0 <= p).sum()) for p in pts))
o = np.argsort(st)
print("grid index in start order (every 200th): %s" % idx[o][::200].tolist())
long_ = idx[dur > np.percentile(dur, 99)]
print("the 1 %% longest-lived workgroups: grid indices %s ..., lives %s" % (long_[:10].tolist(), np.sort(dur)[-10:].tolist()))
