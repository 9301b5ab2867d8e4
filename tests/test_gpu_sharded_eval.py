"""Full-catalogue evaluation over a ROW-SHARDED item table (``evaluate.rank_all`` with ``args.shard_tables``; ``ps_rank_shard``):
top-k lists (catalogue ids, scores) and the target's rank must equal what the unsharded model gives on the same parameters —
``Trainer.test`` / ``calc_metrics`` (trainer.py:125-226) with all products as candidates; the reference keeps the whole table on
one device (item_transformer.py:46), so the split is new and is checked against the single-table path, in one process (world 1)
and as two ranks (gloo, both on cuda:0), each ranking its own batch over both shards.  ``test()`` with explicit candidate
lists goes through the same comparison (rows fetched from their owners in column chunks, then the ordinary score launch)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, 'helpers', 'shard_eval_worker.py')


def _check(pi, ps, pr, si, ss, sr, pc=None, sc=None):
    if pc is not None:                                             # test(): candidate scores, same rows through the same launch
        assert pc.shape == sc.shape == (24, 100)
        assert np.array_equal(pc, sc)
        assert np.abs(pc[:, :-3]).max() > 0
    assert np.array_equal(pr, sr), (pr, sr)                       # the target's rank: bit-exact (index work)
    assert pr[3] == 0 or pr.min() >= 1
    assert np.array_equal(pi, si)                                  # the same products in the same order
    assert np.allclose(ps, ss, rtol=1e-6, atol=1e-6)


def test_one_process_sharded_ranking_equals_the_single_table_ranking():
    sys.path.insert(0, os.path.join(HERE, 'helpers'))
    import shard_eval_worker
    res = shard_eval_worker.run(0, 1)
    _check(*res['plain'][:3], *res['sharded'][:3], res['plain'][3], res['sharded'][3])
    assert res['plain'][2][3] == 0                                 # the row whose target is not a product


@pytest.mark.parametrize('world', [2, 3])      # 3: 6,001 rows in shards of 2,001 / 2,000 / 2,000
def test_ranks_rank_their_batches_over_all_shards(world, tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'shard_eval')
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, WORKER, '--out', out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-4000:])
    for r in range(world):
        z = np.load(out + '.rank%d.npz' % r)
        _check(z['pi'], z['ps'], z['pr'], z['si'], z['ss'], z['sr'], z['pc'], z['sc'])
