import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prodsearch_amd.synth as s
s.make_word_dists = lambda V, seed=7: np.concatenate([np.full(V - 1, 1.0 / (V - 1)), [0.0]])
import bench
sys.argv = ['bench.py', '--steps', '300', '--warmup', '30', '--cpu-steps', '0', '--no-also']
bench.main()
