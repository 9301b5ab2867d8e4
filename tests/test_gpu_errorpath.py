"""A backward that fails between a side-stream fork and its join must surface as an exception, not as a hang.

The default step parks the side stream on ``hipStreamWaitValue32`` and lets the NEXT main-stream kernel store the value
(csrc/tem.hip, side_fork / side_take_signal).  If the call fails before that kernel is launched nobody stores it; the
entry points therefore release the fork on every non-OK exit (``side_abort``).  ``ps_debug_fail_fork(n)`` injects exactly
that failure.  The scenario runs in a child process with a timeout so that a regression cannot stall the whole suite
(trainer.py:74-79 call order)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, torch
sys.path.insert(0, %(root)r)
from prodsearch_amd import ItemTransformerRanker, ProductRanker, _lib, build_optim, default_args, readme_tem_args, synth, rtm_data
lib = _lib.load()
which = sys.argv[1]
if which == 'tem':
    P_, V, B = 3000, 4000, 384
    a = readme_tem_args(dropout=0.1)
    wd = synth.make_word_dists(V)
    m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
    batch = synth.make_tem_batch(1, B, P_, V, word_dists=wd).to('cuda')
    fwd = lambda: m(batch)
else:
    V, RC, B, K, WL = 4000, 5000, 64, 5, 40
    a = default_args(model_name='review_transformer', review_encoder_name='pvc', embedding_size=128, heads=8, ff_size=512,
                     inter_layers=1, neg_per_pos=K, dropout=0.1, corrupt_rate=0.9, review_word_limit=WL,
                     uprev_review_limit=8, iprev_review_limit=8)
    wd = synth.make_word_dists(V)
    rng = synth.rng_for(3)
    rw = torch.from_numpy(rng.integers(0, V - 1, size=(RC, WL)))
    rw[-1] = V - 1
    m = ProductRanker(a, 'cuda', V, RC, 100, 100, rw, None, word_dists=wd)
    batch = rtm_data.make_rtm_batch(7, B, K, RC, V, rw, Q=6, u_lim=8, i_lim=8, W=1, train_pv=False, encoder='pvc',
                                    word_dists=wd).to('cuda')
    fwd = lambda: m(batch, train_pv=False)
optim = build_optim(a, m, None)
m.train()
for _ in range(2):                       # two clean steps (workspaces, plans, side stream exist)
    loss = fwd(); m.zero_grad(); loss.backward(); optim.step()
torch.cuda.synchronize()
raised = 0
for nth in (1, 2):                       # fail behind the first and behind the second fork of a backward
    loss = fwd(); m.zero_grad()
    lib.ps_debug_fail_fork(nth)
    try:
        loss.backward()
    except RuntimeError as e:
        assert 'injected failure' in str(e), e
        raised += 1
    lib.ps_debug_fail_fork(0)
    torch.cuda.synchronize()             # would never return with the side stream still parked
assert raised >= 1, "no fork was reached: the hook did not fire"
loss = fwd(); m.zero_grad(); loss.backward(); optim.step()      # and the model still trains
torch.cuda.synchronize()
assert bool(torch.isfinite(loss.detach()))
print('OK', raised)
'''


@pytest.mark.parametrize('which', ['tem', 'rtm'])
def test_failed_backward_does_not_leave_the_side_stream_waiting(which):
    env = dict(os.environ)
    env.pop('PS_NO_SIDE', None)
    proc = subprocess.Popen([sys.executable, '-c', CHILD % {'root': ROOT}, which], stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True, env=env)
    try:
        out, _ = proc.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        pytest.fail("child hung: a failed backward left the side stream waiting")
    assert proc.returncode == 0 and 'OK' in out, out[-3000:]
