"""Data-parallel step equivalence on the GPU: 2 ranks x B=192 must train exactly like 1 rank x B=384.

Both loss terms are batch means (item_transformer.py:282,514), so the mean of the two ranks' gradients IS the gradient of
the 384-row batch; ``dist.GradExchange`` / ``dist.SparseGradExchange`` sum the ranks' gradients and the fused clip+Adam
applies 1/world (``Optimizer.grad_scale``) before the global-norm clip (optimizers.py:241-243).  The two ranks are fresh
child processes (gloo over 127.0.0.1, both on cuda:0); the single-rank run happens in this process.  Injected negatives,
dropout 0.  Checked after 2 Adam steps, dense and row-sparse modes:
  * the two ranks hold bitwise identical parameters (replicas stay in lock step);
  * they equal the single-rank parameters up to the fp32 reassociation of the gradient sums."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, 'helpers', 'dp_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(mode, out, world=2, steps=2, extra=()):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, WORKER, '--mode', mode, '--steps', str(steps), '--out', out] + list(extra),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-4000:])
    return [dict(np.load(out + '.rank%d.npz' % r)) for r in range(world)]


@pytest.mark.parametrize('mode', ['dense', 'allreduce', 'sparse'])      # dense = sharded optimizer (the default), allreduce = replicated
def test_two_ranks_of_192_equal_one_rank_of_384(mode, tmp_path):
    sys.path.insert(0, os.path.join(HERE, 'helpers'))
    import dp_worker

    class _NoExchange(object):
        def __call__(self):
            return None

    lr, steps = 0.002, 2
    single = dp_worker.run(mode, 384, steps, 0, 1, lambda m, o: _NoExchange())
    ranks = _run_ranks(mode, str(tmp_path / ('dp_' + mode)), 2, steps)
    r0, r1 = ranks
    for k in r0:
        if k.startswith('__'):
            continue
        assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k           # bitwise lock step
    checked = 0
    for k, ref in single.items():
        if k.startswith('__') or k.endswith('__sum') or k.endswith('linear_keys.bias'):
            continue
        got = r0[k]
        assert got.shape == ref.shape, k
        tol = 2e-3 * float(np.abs(ref).max()) + 0.02 * lr * steps
        assert float(np.abs(got - ref).max()) < tol, (k, float(np.abs(got - ref).max()), tol)
        checked += 1
    assert checked >= 20
    for k in ('product_emb.weight__sum', 'word_embeddings.weight__sum'):           # nothing moved outside the compared rows
        if mode == 'sparse':
            assert abs(r0[k] - single[k]) < 1e-3 * max(1.0, abs(single[k])) + 1.0, k
    # per-rank mean loss of the last step: the two halves average to the full batch's loss
    assert abs(0.5 * (r0['__loss'] + r1['__loss']) - single['__loss']) < 2e-3 * abs(single['__loss'])
    print("dp %s: 2-rank step %.3f ms (gloo, one GPU), single-rank %.3f ms" % (mode, r0['__ms'], single['__ms']))


@pytest.mark.parametrize('mode,forms', [('dense', 'rccl'), ('dense', 'a2a'), ('sparse', None), ('sharded', None)])
def test_four_ranks_of_96_equal_one_rank_of_384(mode, forms, tmp_path, monkeypatch):
    """World 4 (the most the one-GPU box lets share its card besides the test process): slice arithmetic of the sharded
    optimizer's flat buffers, the peer-to-peer forms of its collectives with more than one peer, four fixed-capacity row messages,
    four owners of a row-sharded item table — things a world of 2 cannot get wrong."""
    sys.path.insert(0, os.path.join(HERE, 'helpers'))
    import dp_worker

    class _NoExchange(object):
        def __call__(self):
            return None

    if forms:
        monkeypatch.setenv('PS_DP_RS', forms); monkeypatch.setenv('PS_DP_AG', forms)
    lr, steps = 0.002, 2
    single = dp_worker.run('sparse' if mode == 'sharded' else mode, 384, steps, 0, 1, lambda m, o: _NoExchange())
    ranks = _run_ranks(mode, str(tmp_path / ('dp4_' + mode)), 4, steps)
    for r in ranks[1:]:
        for k in ranks[0]:
            if not k.startswith('__'):
                assert np.array_equal(ranks[0][k], r[k]), "replicas diverged: " + k
    checked = 0
    for k, ref in single.items():
        if k.startswith('__') or k.endswith('__sum') or k.endswith('linear_keys.bias'):
            continue
        got = ranks[0][k]
        assert got.shape == ref.shape, k
        tol = 2e-3 * float(np.abs(ref).max()) + 0.02 * lr * steps
        assert float(np.abs(got - ref).max()) < tol, (k, float(np.abs(got - ref).max()), tol)
        checked += 1
    assert checked >= 20
    assert abs(sum(r['__loss'] for r in ranks) / 4 - single['__loss']) < 2e-3 * abs(single['__loss'])


@pytest.mark.parametrize('mode', ['dense', 'sparse'])
def test_unequal_per_rank_batches_keep_the_replicas_in_lock_step(mode, tmp_path):
    """Ranks whose batches differ in size (192 and 150 rows: a loader with drop_last=False) address different numbers of
    table rows; the row-sparse exchange's fixed-capacity messages are sized by the capacity the ranks agree on at the first
    step (dist.SparseGradExchange._agree_capacity), so the collectives match and the replicas stay bitwise identical."""
    r0, r1 = _run_ranks(mode, str(tmp_path / ('rag_' + mode)), 2, 2, extra=('--ragged', '42'))
    for k in r0:
        if not k.startswith('__'):
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k
    assert np.isfinite(r0['__loss']) and np.isfinite(r1['__loss'])


def test_two_ranks_with_lazy_exact_adam_equal_one_dense_rank(tmp_path):
    """``args.lazy_exact_adam`` under data parallelism: the row-sparse exchange widens the touched lists to the ranks' union after
    the backward, and the optimizer's replay runs over THOSE lists before the step, so rows only the other rank addressed are
    brought up to date too; three steps of two ranks must equal three steps of the single-process DENSE optimizer."""
    sys.path.insert(0, os.path.join(HERE, 'helpers'))
    import dp_worker

    class _NoExchange(object):
        def __call__(self):
            return None

    lr, steps = 0.002, 3
    single = dp_worker.run('dense', 384, steps, 0, 1, lambda m, o: _NoExchange())
    r0, r1 = _run_ranks('lazy', str(tmp_path / 'dp_lazy'), 2, steps)
    for k in r0:
        if not k.startswith('__'):
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k
    checked = 0
    for k, ref in single.items():
        if k.startswith('__') or k.endswith('linear_keys.bias'):
            continue
        got = r0[k]
        if k.endswith('__sum'):
            assert abs(got - ref) < 1e-3 * max(1.0, abs(ref)) + 1.0, k          # every row of the table, flushed
            continue
        assert got.shape == ref.shape, k
        tol = 2e-3 * float(np.abs(ref).max()) + 0.02 * lr * steps
        assert float(np.abs(got - ref).max()) < tol, (k, float(np.abs(got - ref).max()), tol)
        checked += 1
    assert checked >= 20


def test_timed_choice_of_the_collective_forms_changes_no_result(tmp_path, monkeypatch):
    """dist.ShardedAdamExchange times both forms of its reduce-scatter and all-gather (the library's / slices sent peer to peer
    with all_to_all_single) inside the first exchange and keeps the faster; on RCCL that is the default, here it is forced on
    over gloo.  Whatever it picks, two ranks' sums are the same two addends: parameters equal the untimed run bit for bit."""
    monkeypatch.setenv('PS_DETERMINISTIC', '1')          # (the default step's fp32 atomics differ from run to run by themselves)
    monkeypatch.setenv('PS_DP_RS', 'rccl'); monkeypatch.setenv('PS_DP_AG', 'rccl')
    base = _run_ranks('dense', str(tmp_path / 'tune_base'), 2, 2)
    monkeypatch.setenv('PS_DP_RS', 'auto'); monkeypatch.setenv('PS_DP_AG', 'auto')
    tuned = _run_ranks('dense', str(tmp_path / 'tune_auto'), 2, 2)
    monkeypatch.setenv('PS_DP_RS', 'a2a'); monkeypatch.setenv('PS_DP_AG', 'a2a')
    a2a = _run_ranks('dense', str(tmp_path / 'tune_a2a'), 2, 2)
    for other in (tuned, a2a):
        for r in range(2):
            for k in base[r]:
                if not k.startswith('__'):
                    assert np.array_equal(base[r][k], other[r][k]), k


def test_sparse_exchange_is_not_slower_than_twice_the_dense_one(tmp_path):
    """The device-side merge (ps_pack_rows / ps_merge_rows) keeps the row-sparse exchange free of host syncs: as a 2-rank
    gloo dry run on one GPU its step must stay within 2x of the dense exchange's (it was 59 ms against 1 ms with the
    torch.unique / searchsorted glue it replaces)."""
    dense = _run_ranks('dense', str(tmp_path / 'td'), 2, 8)
    sparse = _run_ranks('sparse', str(tmp_path / 'ts'), 2, 8)
    t_d, t_s = float(dense[0]['__ms']), float(sparse[0]['__ms'])
    print("2-rank gloo dry run, ms/step: dense %.3f, sparse %.3f" % (t_d, t_s))
    assert t_s < 2.0 * t_d + 0.5


def test_review_transformer_two_ranks_of_32_equal_one_rank_of_64(tmp_path):
    """The same for the review transformer (dense exchange): its step carries state across kernels that the TEM step does
    not have — the word counts / ranks of the inverted index, the loss ticket, the parked segment-embedding partials."""
    sys.path.insert(0, os.path.join(HERE, 'helpers'))
    import dp_worker

    class _NoExchange(object):
        def __call__(self):
            return None

    lr, steps = 0.002, 2
    single = dp_worker.run_rtm(64, steps, 0, 1, lambda m, o: _NoExchange())
    r0, r1 = _run_ranks('dense', str(tmp_path / 'dp_rtm'), 2, steps, extra=('--model', 'rtm', '--global-batch', '64'))
    for k in r0:
        if not k.startswith('__'):
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k       # bitwise lock step
    checked = 0
    for k, ref in single.items():
        if k.startswith('__') or k.endswith('linear_keys.bias') or float(np.abs(ref).max()) == 0.0:
            continue
        got = r0[k]
        tol = 2e-3 * float(np.abs(ref).max()) + 0.02 * lr * steps
        assert float(np.abs(got - ref).max()) < tol, (k, float(np.abs(got - ref).max()), tol)
        checked += 1
    assert checked >= 18
    assert abs(0.5 * (r0['__loss'] + r1['__loss']) - single['__loss']) < 2e-3 * abs(single['__loss'])


def test_bench_contract_line_from_two_ranks_with_its_extra_legs():
    """The N>1 launch of bench.py exactly as the driver starts it (one process per rank, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* from the environment; gloo here, both ranks on cuda:0), with the median / roofline legs on: they contain the
    gradient exchange, so every rank has to run them — rank 0 alone would wait in its all-reduce for ever.  One JSON line,
    from rank 0 only, whole-job value = 2 ranks' tuples."""
    import json
    port = _free_port()
    bench = os.path.join(os.path.dirname(HERE), 'bench.py')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0', PS_DIST_BACKEND='gloo', PS_BENCH_WATCHDOG='300')
        procs.append(subprocess.Popen([sys.executable, bench, '--gpus', '2', '--steps', '6', '--warmup', '2', '--reps', '8'],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=360)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    lines0 = [ln for ln in outs[0].splitlines() if ln.startswith('{')]
    assert len(lines0) == 1 and not [ln for ln in outs[1].splitlines() if ln.startswith('{')]
    d = json.loads(lines0[0])
    assert d['n_gpus'] == 2 and d['steps'] == 6 and d['scaling'] == 'weak' and d['config']['parallelism'] == 'dp2'
    assert abs(d['value'] - 2 * 384 * 20 * 6 / (d['ms_per_step'] * 6e-3)) < 1e-6 * d['value']
    assert d['roofline']['launches_timed'] == 6 and 'median_ms_per_step' in d and 'cpu_baseline' not in d


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with no torchrun environment starts the two ranks itself, before anything touches the GPU
    (gloo here: both ranks share cuda:0), and prints ONE line with n_gpus = 2; with one visible device and the default
    backend (RCCL: one rank per GPU) it must exit non-zero and print NO line — never a silent world-1 run."""
    import json
    bench = os.path.join(os.path.dirname(HERE), 'bench.py')
    env = dict(os.environ, PS_DIST_BACKEND='gloo', PS_BENCH_WATCHDOG='300')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '6', '--warmup', '2', '--reps', '0', '--cpu-steps', '0'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 768
    assert 'reduce-scatter' in d['config']['step']                     # the sharded optimizer is the dense default
    if torch.cuda.device_count() < 2:
        env.pop('PS_DIST_BACKEND')
        p = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '2', '--warmup', '0', '--no-extras'], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    env['WORLD_SIZE'], env['RANK'], env['LOCAL_RANK'] = '1', '0', '0'     # a launcher that started ONE rank for --gpus 2
    p = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '2', '--warmup', '0', '--no-extras'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]


def test_bench_self_launch_falls_back_to_the_library_collectives_once():
    """the sharded optimizer's peer-to-peer forms meet RCCL for the first time on the driver's node.  If the first
    set of self-launched ranks fails, bench.py starts ONE fresh set of child processes with PS_DP_RS=rccl PS_DP_AG=rccl and says
    so in the line.  Here rank 1 of the first set dies before touching the GPU (PS_BENCH_TEST_FAIL_P2P), rank 0 is left in
    the rendezvous and must be killed, and the second set (gloo, both ranks on cuda:0) delivers the line."""
    import json
    bench = os.path.join(os.path.dirname(HERE), 'bench.py')
    env = dict(os.environ, PS_DIST_BACKEND='gloo', PS_BENCH_WATCHDOG='300', PS_BENCH_TEST_FAIL_P2P='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'PS_DP_RS', 'PS_DP_AG'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '4', '--warmup', '1', '--no-extras'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d.get('exchange_fallback') is True
    assert d['exchange_fallback_detail']['first_attempt_exit_codes'][1] == 3
    assert d['exchange_fallback_detail']['env'] == {'PS_DP_RS': 'rccl', 'PS_DP_AG': 'rccl'}
    # with the library collectives pinned from the start there is nothing to fall back to: the failure is final
    env.update(PS_DP_RS='rccl', PS_DP_AG='rccl')
    env.pop('PS_BENCH_TEST_FAIL_P2P')
    p = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '2', '--warmup', '0', '--no-extras'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0 and 'exchange_fallback' not in json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][0])


def test_side_stream_sequence_words_restart_before_they_wrap():
    """Forks and joins of the side stream compare 32-bit sequence words with >=; side_fork drains both streams and restarts
    the words long before they wrap.  PS_SIDE_SEQ0 starts them just under that threshold: the same 60 steps must give the
    loss of a run that starts from zero (the guard is crossed about 20 steps in)."""
    import json
    bench = os.path.join(os.path.dirname(HERE), 'bench.py')
    losses = []
    for seq0 in (None, '0xffefffd8'):
        env = dict(os.environ, PS_BENCH_WATCHDOG='240')
        env.pop('PS_SIDE_SEQ0', None)
        if seq0:
            env['PS_SIDE_SEQ0'] = seq0
        p = subprocess.run([sys.executable, bench, '--steps', '60', '--warmup', '0', '--no-extras'], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
        losses.append(d['final_loss'])
    assert abs(losses[0] - losses[1]) <= 1e-4 * abs(losses[0]), losses
