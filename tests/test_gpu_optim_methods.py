"""The reference's other --optim methods (models/optimizers.py:175-183: sgd, adagrad, adadelta) through the same two launches as
Adam: clip by the global norm, then torch.optim's update rule with its defaults.  Checked against torch's own optimizers run on
the CPU in fp32 on the same gradients (tolerance: a few ulps of fp32 — the only difference is fused vs unfused multiply-adds)."""
import pytest
import torch
from torch.nn.utils import clip_grad_norm_

from prodsearch_amd.optimizers import Optimizer

pytestmark = pytest.mark.gpu
SHAPES = [(300, 128), (128,), (5000,), (33, 7), (4096,), (1,)]


def _ref_optimizer(method, params, lr, wd, accum):
    if method == 'sgd':
        return torch.optim.SGD(params, lr=lr, weight_decay=wd)
    if method == 'adagrad':
        o = torch.optim.Adagrad(params, lr=lr, weight_decay=wd)
        for g in o.param_groups:
            for p in g['params']:
                o.state[p]['sum'] = o.state[p]['sum'].fill_(accum)                  # optimizers.py:178-181
        return o
    if method == 'adadelta':
        return torch.optim.Adadelta(params, lr=lr, weight_decay=wd)
    raise ValueError(method)         # (Adam: tests/test_gpu_parity.py against the reference's own fixtures)


@pytest.mark.parametrize('method', ['sgd', 'adagrad', 'adadelta'])
@pytest.mark.parametrize('noam,wd,clip', [(False, 0.0, 5.0), (True, 0.01, 0.5), (False, 0.02, 0.0)])
def test_methods_match_torch_optim(method, noam, wd, clip):
    torch.manual_seed(3)
    lr, accum, warm = (0.05 if method != 'adadelta' else 1.0), 0.1, 4
    ref = [torch.nn.Parameter(torch.randn(s)) for s in SHAPES]
    dev = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    ro = _ref_optimizer(method, ref, lr, wd, accum)
    opt = Optimizer(method, lr, clip, decay_method='noam' if noam else None, warmup_steps=warm, weight_decay=wd, adagrad_accum=accum)
    opt.set_parameters([('p%d' % i, p) for i, p in enumerate(dev)])
    for step in range(1, 7):
        for p, q in zip(ref, dev):
            g = torch.randn(p.shape) * (3.0 if step % 2 else 0.3)
            if p.numel() == 5000:
                g[1000:3000] = 0.0                               # zero runs: the update's non-zero mask
            p.grad = g.clone()
            q.grad = g.cuda()
        if noam:                                                 # optimizers.py:214-219, :236
            for grp in ro.param_groups:
                grp['lr'] = lr * min(step ** -0.5, step * warm ** -1.5)
        if clip:
            clip_grad_norm_(ref, clip)
        ro.step()
        opt.step()
        for p, q in zip(ref, dev):
            assert torch.allclose(q.detach().cpu(), p.detach(), rtol=2e-6, atol=2e-7), (method, step, tuple(p.shape))
    if noam:
        assert abs(opt.learning_rate - lr * min(6 ** -0.5, 6 * warm ** -1.5)) < 1e-9


@pytest.mark.parametrize('method', ['sgd', 'adagrad', 'adadelta'])
def test_state_dict_round_trip_continues_the_run(method):
    torch.manual_seed(4)
    mk = lambda: [torch.nn.Parameter(torch.randn(s).cuda()) for s in SHAPES[:3]]
    a, grads = mk(), [[torch.randn(s).cuda() for s in SHAPES[:3]] for _ in range(4)]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]

    def new(ps):
        o = Optimizer(method, 0.1, 1.0, adagrad_accum=0.05)
        o.set_parameters([('p%d' % i, p) for i, p in enumerate(ps)])
        return o
    oa, ob = new(a), new(b)
    for k in range(4):
        if k == 2:                                               # save / load into a fresh optimizer half way
            sd = ob.state_dict()
            assert set(k2 for s in sd['state'].values() for k2 in s) == set(Optimizer.STATE_KEYS[method]) | ({'step'} if method != 'sgd' else set())
            ob = new(b)
            ob.load_state_dict(sd)
        for ps, o in ((a, oa), (b, ob)):
            for p, g in zip(ps, grads[k]):
                p.grad = g.clone()
            o.step()
    for p, q in zip(a, b):
        assert torch.equal(p, q)


def test_unknown_methods_raise():
    with pytest.raises(RuntimeError, match='Invalid optim method'):
        Optimizer('rmsprop', 0.1, 1.0)
    with pytest.raises(NotImplementedError):
        Optimizer('sparseadam', 0.1, 1.0)
    with pytest.raises(NotImplementedError):
        Optimizer('sgd', 0.1, 1.0, row_sparse=True)


@pytest.mark.parametrize('method', ['sgd', 'adagrad'])
def test_model_trains_with_the_other_methods(method):
    """--optim sgd / adagrad through build_optim on the item transformer: every step equals torch's optimizer applied to the
    gradients the step produced (the model's flat gradient buffer is cleared by the update, like under Adam)."""
    from prodsearch_amd import ItemTransformerRanker, build_optim, default_args, synth
    args = default_args(model_name='item_transformer', embedding_size=64, ff_size=128, heads=4, inter_layers=1, neg_per_pos=5,
                        dropout=0.0, optim=method, lr=0.05, max_grad_norm=1.0, decay_method='adam')
    V, P = 300, 200
    wd = synth.make_word_dists(V)
    torch.manual_seed(0)
    m = ItemTransformerRanker(args, 'cuda', V, P, None, word_dists=wd)
    opt = build_optim(args, m, None)
    m.train()
    names = [k for k, p in m.named_parameters() if p.requires_grad]
    ref = {k: torch.nn.Parameter(p.detach().cpu().clone()) for k, p in m.named_parameters() if p.requires_grad}
    ro = _ref_optimizer(method, list(ref.values()), 0.05, 0.0, 0.0)
    for step in range(3):
        b = synth.make_tem_batch(50 + step, 16, P, V, Q=4, L=6, W=2, word_dists=wd).to('cuda')
        loss = m(b)
        m.zero_grad()
        loss.backward()
        live = []
        for k, p in m.named_parameters():
            if p.requires_grad and p.grad is not None:
                ref[k].grad = p.grad.detach().cpu().clone()
                live.append(ref[k])
        clip_grad_norm_(live, 1.0)
        ro.step()
        opt.step()
        for k, p in m.named_parameters():
            if p.requires_grad and p.grad is not None:
                assert torch.allclose(p.detach().cpu(), ref[k].detach(), rtol=3e-6, atol=3e-7), (method, step, k)
                assert float(p.grad.abs().max()) == 0.0          # consumed and cleared by the update
    assert names
