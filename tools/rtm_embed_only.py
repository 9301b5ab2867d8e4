"""Forward-only loop of the review transformer at BASELINE configs[3]: target for rocprofv3 --pmc passes over rtm_embed4_kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch
import bench
a = argparse.Namespace(workload='c4', encoder='pvc', dropout=0.1, row_sparse=False)
wl = bench.RtmWorkload(a, 'c4', 0, torch.device('cuda', 0))
wl.model.train()
with torch.no_grad():
    for i in range(30):
        wl.forward(i)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", wl.roofline_spec()['work'])
