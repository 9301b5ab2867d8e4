import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from golden_util import rel_err
from golden_util_rtm import RtmGolden
import test_gpu_parity_rtm as T
for case in ['rtm_fs', 'rtm_avg']:
    g = RtmGolden(case)
    m = T._model(g)
    b = g.batch().to('cuda')
    loss = m(b, train_pv=False)
    m.zero_grad(); loss.backward(); torch.cuda.synchronize()
    print(case, 'loss', float(loss), float(g.tensor('loss_0')))
    for n, p in m.named_parameters():
        if p.grad is None: continue
        ref = g.tensor('grad_' + n)
        e = rel_err(p.grad.cpu(), ref)
        print('  %-70s %.2e' % (n, e))
    if case == 'rtm_fs':
        ge, re_ = m.word_embeddings.weight.grad.cpu(), g.tensor('grad_word_embeddings.weight')
        diff = (ge - re_).abs().amax(1)
        bad = torch.nonzero(diff > 1e-3 * re_.abs().max()).flatten()
        print('bad rows', bad.tolist()[:20], 'of', int((re_.abs().amax(1) > 0).sum()))
        qw = set(g.batch().query_word_idxs.flatten().tolist())
        print('bad rows that are query words:', [int(x) for x in bad if int(x) in qw][:20])
        masks_equal = bool((g.batch().pos_prod_rword_masks.bool() == (g.batch().pos_prod_rword_idxs != g.V - 1)).all())
        print('pos masks == (idx != pad):', masks_equal)
