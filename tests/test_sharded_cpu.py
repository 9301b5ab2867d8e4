"""Row-sharded item table (prodsearch_amd/sharded.py) on CPU: world_size-2 gloo processes drive the
fixed-capacity all-to-all protocol of ``ShardedItemTable`` — with the five HIP kernels it calls restated in torch, so the
collectives, capacities, slot arithmetic and the rank-ordered merge are what is under test — against a replicated table:
the receive buffer holds exactly the rows a rank's batch addresses at the slots its remapped indices name, the gradients
routed back sum, in every owner's shard, to the gradient a replicated table would have received from both ranks, and an
overflowing request sets the status word instead of corrupting anything.  (The reference has no counterpart: it keeps the
whole table on one device, item_transformer.py:46,464-469.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from prodsearch_amd import dist as pdist
from prodsearch_amd.sharded import ShardedItemTable


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class CpuShard(ShardedItemTable):
    """The kernels' contracts (include/prodsearch_hip.h) in torch."""

    def _k_coalesce(self, tensors, n_rows, pad, ws, rows, cap, count):
        flat = torch.cat([t.reshape(-1) for t in tensors])
        u = torch.unique(flat[flat != pad])
        assert u.numel() <= cap and (u.numel() == 0 or (int(u[0]) >= 0 and int(u[-1]) < n_rows))
        rows[:u.numel()] = u
        count[0] = u.numel()

    def _k_bucket(self):
        n, W, capp = int(self.count[0]), self.world, self.capp
        self.send_ids.fill_(-1)
        pos = [0] * W
        for i in range(n):
            r = int(self.rows[i]); o = r % W
            if pos[o] < capp:
                self.send_ids[o, pos[o]] = r // W
                self.slot_of[i] = o * capp + pos[o]
            else:
                self.bad[0] = 2
                self.slot_of[i] = W * capp
            pos[o] += 1

    def _k_gather(self, out):
        a = self.asked.view(-1)
        live = a >= 0
        out[:self.slots][live] = self.weight[a[live]]
        out[:self.slots][~live] = 0

    def _k_remap(self, t, out):
        n = int(self.count[0])
        rows = self.rows[:n]
        flat = t.reshape(-1)
        pos = torch.searchsorted(rows, flat.clamp(min=0)) if n else torch.zeros_like(flat)
        pos = pos.clamp(max=max(n - 1, 0))
        found = (flat != self.pad_row) & (n > 0) & (rows[pos] == flat if n else torch.zeros_like(flat, dtype=torch.bool))
        out.copy_(torch.where(found, self.slot_of[pos].long() if n else torch.zeros_like(flat), torch.full_like(flat, self.slots)))
        if bool(((flat != self.pad_row) & ~found).any()):
            self.bad[0] = 1

    def _k_merge(self, got):
        n = int(self.ucount[0])
        for u in range(n):
            row = int(self.urows[u])
            acc = torch.zeros(self.d)
            for r in range(self.world):                      # rank order: bitwise the same on every run
                lst = self.asked[r]
                hit = (lst == row).nonzero()
                if hit.numel():
                    acc = acc + got[r * self.capp + int(hit[0])]
            self.grad[row] = acc


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    pdist.init_from_env(backend='gloo')
    n_rows, d, pad, cap = 1001, 8, 1001, 600
    full = torch.randn(n_rows, d, generator=torch.Generator().manual_seed(5))            # identical on every rank
    tab = CpuShard(n_rows, d, pad, cap, 'cpu')
    tab.load_full(full)
    ok = [bool(torch.equal(tab.gather_full(), full)), tab.capp < cap and tab.slots == world * tab.capp]
    gen = torch.Generator().manual_seed(100 + rank)
    ref_grad = torch.zeros(n_rows, d)
    for step in range(3):
        # ragged per-rank batches with shared rows, repeats and padding entries; rank 1's last step addresses nothing
        n = (37, 90)[rank] if not (rank == 1 and step == 2) else 0
        tgt = torch.randint(0, n_rows, (n,), generator=gen)
        hist = torch.randint(0, n_rows, (n, 5), generator=gen)
        hist[torch.rand(n, 5, generator=gen) < 0.3] = pad
        rt, rh = tab.lookup([tgt, hist])
        buf = tab.table_buf
        # every remapped index addresses the row the original index names; padding goes to the zero row
        want_t = full[tgt]
        want_h = torch.where((hist == pad)[..., None], torch.zeros(1, d), full[hist.clamp(max=n_rows - 1)])
        ok.append(bool(torch.equal(buf[rt], want_t)) and bool(torch.equal(buf[rh], want_h)))
        ok.append(bool((rh[hist == pad] == tab.slots).all()) and float(buf[tab.slots].abs().sum()) == 0.0)
        # a gradient per addressed slot: scatter-add of per-occurrence rows, as the step's kernels do
        g = torch.zeros(tab.slots + 1, d)
        occ_t = torch.randn(n, d, generator=gen)
        occ_h = torch.randn(n, 5, d, generator=gen)
        g.index_add_(0, rt, occ_t)
        g.index_add_(0, rh.reshape(-1), occ_h.reshape(-1, d))
        g[tab.slots] = 0
        tab.grad.zero_()
        tab.push_grads(g)
        # replicated reference: both ranks' occurrences summed into a dense [n_rows, d] gradient
        mine = torch.zeros(n_rows + 1, d)
        mine.index_add_(0, tgt, occ_t)
        mine.index_add_(0, hist.reshape(-1), occ_h.reshape(-1, d))
        mine = mine[:n_rows].contiguous()
        dist.all_reduce(mine)
        got_full = tab.gather_full('grad')
        ok.append(bool(torch.allclose(got_full, mine, atol=1e-5)))
        touched = set(tab.urows[:int(tab.ucount[0])].tolist())
        want_touched = set((torch.nonzero(mine.ne(0).any(1)).flatten()[rank::1]).tolist())
        want_local = {r // world for r in want_touched if r % world == rank}
        ok.append(want_local <= touched)
        ok.append(int(tab.bad[0]) == 0)
    # a skewed step (every id owned by rank 0) overflows the per-owner capacity: flagged, never silently wrong
    skew = torch.arange(0, 2 * (tab.capp + 5), 2)[:tab.capp + 5] if world == 2 else torch.arange(tab.capp + 5)
    tab.lookup([skew.clamp(max=n_rows - 1)])
    flagged = int(tab.bad[0]) == 2
    raised = False
    try:
        tab.check_errors()
    except RuntimeError:
        raised = True
    q.put((rank, ok, flagged or tab.capp + 5 > n_rows // 2, raised or not flagged))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_sharded_item_table_protocol():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, flagged, raised in res:
        assert all(ok), (rank, ok)
        assert flagged and raised, (rank, flagged, raised)


def test_single_process_round_trip():
    n_rows, d, pad = 200, 4, 200
    full = torch.randn(n_rows, d, generator=torch.Generator().manual_seed(1))
    tab = CpuShard(n_rows, d, pad, 64, 'cpu')
    tab.load_full(full)
    idx = torch.tensor([5, 7, 7, pad, 199, 0])
    (r,) = tab.lookup([idx])
    assert torch.equal(tab.table_buf[r][[0, 1, 2, 4, 5]], full[idx[[0, 1, 2, 4, 5]]]) and int(r[3]) == tab.slots
    g = torch.zeros(tab.slots + 1, d)
    g.index_add_(0, r, torch.ones(6, d))
    g[tab.slots] = 0
    tab.push_grads(g)
    want = torch.zeros(n_rows, d)
    want.index_add_(0, idx[idx != pad], torch.ones(5, d))
    assert torch.equal(tab.gather_full('grad'), want)
