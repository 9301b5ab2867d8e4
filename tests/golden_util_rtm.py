"""Load an RTM golden fixture (tests/golden/rtm_*.npz, written by make_golden_rtm.py from the reference)."""
import json
import os

import numpy as np
import torch

from prodsearch_amd import synth, rtm_data
from prodsearch_amd.config import default_args

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
RTM_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith('rtm_') and f.endswith('.npz'))


class RtmGolden(object):
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
        m = self.meta = json.loads(str(self.z['meta']))
        self.args = default_args(**m['args'])
        self.args.device = 'cpu'
        self.args.do_subsample_mask = True
        self.args.review_word_limit = m['WL']
        self.V, self.RC, self.B, self.K, self.R, self.W = m['V'], m['RC'], m['B'], m['K'], m['R'], m['W']
        self.steps, self.train_pv = m['steps'], m['train_pv']
        self.word_dists = self.z['in_word_dists']
        self.review_words = torch.from_numpy(self.z['in_review_words'])

    def params(self):
        shapes = {k: tuple(v) for k, v in self.meta['param_shapes'].items()}
        sd = synth.make_state_dict(shapes, self.meta['weight_seed'], {})
        for k, v in sd.items():
            assert synth.checksum(v) == self.meta['weight_checksum'][k], "weight generator drifted: " + k
        return sd

    def batch(self):
        vals = []
        for k in rtm_data._TRAIN_FIELDS:
            key = 'in_' + k
            vals.append(torch.from_numpy(self.z[key]) if key in self.z.files else None)
        return rtm_data.ProdSearchTrainBatch(*vals, to_tensor=False)

    def test_batch(self):
        t = lambda k: torch.from_numpy(self.z['in_test_' + k]) if 'in_test_' + k in self.z.files else None
        B = self.B
        return rtm_data.ProdSearchTestBatch(list(range(B)), list(range(B)), None, None, t('query_word_idxs'),
                                            t('candi_prod_ridxs'), t('candi_seg_idxs'), t('candi_seq_user_idxs'),
                                            t('candi_seq_item_idxs'), to_tensor=False)

    def neg_words(self, step):
        k = 'in_neg_word_idxs_%d' % step
        return torch.from_numpy(self.z[k]) if k in self.z.files else None

    def dropout(self, step):
        from oracle.philox import RtmPhiloxDropout
        a = self.args
        pvc = a.review_encoder_name == 'pvc'
        if a.dropout <= 0 and not (pvc and a.corrupt_rate > 0):
            return None
        return RtmPhiloxDropout(a.dropout, a.seed, step + 1, self.B, self.K, a.heads, self.R + 1, a.inter_layers,
                                a.corrupt_rate if pvc else 0.0)

    def tensor(self, key, base=None):
        if key in self.z.files:
            return torch.from_numpy(self.z[key])
        rows = self.z[key + '__rows']
        shape = tuple(self.z[key + '__shape'])
        full = torch.zeros(shape) if base is None else base.clone()
        full[torch.from_numpy(rows)] = torch.from_numpy(self.z[key + '__vals'])
        return full
