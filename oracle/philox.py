"""Oracle (TEST INFRASTRUCTURE): the product's counter-based dropout masks, restated.

The HIP kernels draw dropout from Philox4x32-10 (Salmon et al., SC'11 — the
generator torch/curand use) keyed by ``(seed, step)``; element ``(row, col)`` of
dropout site ``s`` uses counter ``(col, row >> 2, s, step)`` and output word
``row & 3``; it is kept iff ``word >= floor(p * 2**32)`` and scaled by ``1/(1-p)``
— or, for the ctx / ff1 / ff2 sites, the 16-bit half ``col & 7`` of counter
``(col >> 3, row, s, step)`` (``drop_mult`` below; prodsearch_amd/csrc/common.h).  This file restates that function in numpy so
the oracle — and, through the torch dropout hooks of tests/golden/make_golden.py,
the REFERENCE itself — can run with exactly the masks the kernels draw.

Row/col conventions per site (they follow the kernels' replica-row layout):
  fs   : [B, d]            row = b,                         col = channel
  attn : [N, H, Sq, S]     row = (nout*H + h)*Sq + i,       col = key
  ctx  : [N, Sq, d]        row = nout*Sq + i,               col = channel
  ff1  : [N, Sq, F]        row = nout*Sq + i,               col = unit
  ff2  : [N, Sq, d]        row = nout*Sq + i,               col = channel
with ``nout = b*R + j`` (j = 0 positive, j = 1+k negative k, R = K+1) and, in the
LAST layer, ``Sq = 1`` (only the consumed position ``qpos`` exists; every other
position's mask cannot influence the result and is set to "keep").
"""
import numpy as np
import torch

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)
SH = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10; inputs broadcastable uint64 arrays holding 32-bit values."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & MASK for x in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0) & MASK, np.uint64(k1) & MASK
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> SH) ^ c1 ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> SH) ^ c3 ^ k1) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def site_is_half(site):
    """Sites ctx / ff1 / ff2 of every layer take the 16-bit column-shared form (csrc/common.h, drop_site_is_half)."""
    k = (site - 1) & 7
    return 1 <= site < 0x100 and 1 <= k <= 3


def drop_mult(rows, cols, site, step, seed, p):
    """Multiplier array (0 or 1/(1-p), float32) for broadcastable integer arrays rows/cols.

    classic form: word ``row & 3`` of Philox(col, row >> 2, site, step), kept iff >= floor(p * 2**32);
    half form (ctx / ff1 / ff2): 16-bit half ``col & 7`` of Philox(col >> 3, row, site, step) — low half of word
    ``(col & 7) >> 1`` for even ``col``, high half for odd — kept iff >= floor(p * 2**16)."""
    rows = np.asarray(rows, dtype=np.uint64)
    cols = np.asarray(cols, dtype=np.uint64)
    rows, cols = np.broadcast_arrays(rows, cols)
    scale = np.float32(1.0 / (1.0 - float(np.float32(p))))
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    if site_is_half(int(site)):
        thr16 = np.uint64(int(float(np.float32(p)) * 65536.0))
        w = philox4x32_10(cols >> np.uint64(3), rows, np.uint64(site), np.uint64(step & 0xFFFFFFFF), k0, k1)
        e = (cols & np.uint64(7)).astype(np.int64)
        word = np.choose(e >> 1, w)
        hv = np.where((e & 1) == 1, word >> np.uint64(16), word & np.uint64(0xFFFF))
        return np.where(hv >= thr16, scale, np.float32(0)).astype(np.float32)
    thr = np.uint64(int(float(np.float32(p)) * 4294967296.0))
    w = philox4x32_10(cols, rows >> np.uint64(2), np.uint64(site), np.uint64(step & 0xFFFFFFFF), k0, k1)
    sel = (rows & np.uint64(3)).astype(np.int64)
    word = np.choose(sel, w)
    return np.where(word >= thr, scale, np.float32(0)).astype(np.float32)


SITE_FS = 0


def site_id(kind, layer):
    return 1 + 8 * layer + {'attn': 0, 'ctx': 1, 'ff1': 2, 'ff2': 3}[kind]


class PhiloxDropout(object):
    """Dropout callable for oracle.tem (``drop(x, kind, call)``) in ``replicate=True``
    structure: ``call`` is 0 (fs), or (c, layer) with c = 0 for the positive encode
    (N = B sequences) and c = 1 for the negatives (N = B*K)."""

    def __init__(self, p, seed, step, B, K, H, S, n_layers, qpos):
        self.p, self.seed, self.step = float(p), int(seed), int(step)
        self.B, self.K, self.H, self.S, self.NL, self.qpos = B, K, H, S, n_layers, qpos
        self.R = K + 1

    def _nout(self, c, N):
        n = np.arange(N)
        if c == 0:
            return n * self.R                       # b*R + 0
        return (n // self.K) * self.R + 1 + (n % self.K)

    def mult(self, shape, kind, call):
        if kind == 'fs':
            B, d = shape
            return drop_mult(np.arange(B)[:, None], np.arange(d)[None, :], SITE_FS, self.step, self.seed, self.p)
        c, layer = call
        last = layer == self.NL - 1
        site = site_id(kind, layer)
        N = shape[0]
        nout = self._nout(c, N)
        S = self.S
        if kind == 'attn':
            _, H, Sq, Sk = shape
            h = np.arange(H)[None, :, None, None]
            i = np.arange(Sq)[None, None, :, None]
            s = np.arange(Sk)[None, None, None, :]
            no = nout[:, None, None, None]
            if last:
                m = np.ones(shape, dtype=np.float32)
                rows = (no * H + h) * 1 + 0 * i[:, :, :1]
                m[:, :, self.qpos % S:self.qpos % S + 1, :] = drop_mult(rows, s, site, self.step, self.seed, self.p)
                return m
            return drop_mult((no * H + h) * Sq + i, s, site, self.step, self.seed, self.p)
        _, Sq, width = shape
        i = np.arange(Sq)[None, :, None]
        col = np.arange(width)[None, None, :]
        no = nout[:, None, None]
        if last:
            m = np.ones(shape, dtype=np.float32)
            q = self.qpos % S
            m[:, q:q + 1, :] = drop_mult(no + 0 * i[:, :1], col, site, self.step, self.seed, self.p)
            return m
        return drop_mult(no * Sq + i, col, site, self.step, self.seed, self.p)

    def __call__(self, x, kind, call):
        if self.p == 0.0:
            return x
        return x * torch.from_numpy(self.mult(tuple(x.shape), kind, call))


# ----------------------------------------------------------------------------- RTM
SITE_REV_PV, SITE_REV_POS, SITE_REV_NEG, SITE_TOK_POS, SITE_TOK_NEG = 0x200, 0x201, 0x202, 0x203, 0x204


class RtmPhiloxDropout(PhiloxDropout):
    """Masks of the RTM step (oracle.rtm): the encoder sites of the parent class (sequence
    n = b*(1+K)+j, so ``R = K+1`` rows per batch row and no replica fan-out) plus the review
    sites:  rev_pv / rev_pos [B*R,d] row = b*R+r;  rev_neg [B,K,R,d] row = (b*K+k)*R+r;
    token dropout of the pvc encoder [N,WL] row = review row, col = word slot, p = corrupt_rate."""

    def __init__(self, p, seed, step, B, K, H, S, n_layers, corrupt_rate=0.0):
        super().__init__(p, seed, step, B, K, H, S, n_layers, 0)
        self.corrupt_rate = float(corrupt_rate)

    def mult(self, shape, kind, call):
        if kind in ('rev_pv', 'rev_pos', 'rev_neg'):
            site = {'rev_pv': SITE_REV_PV, 'rev_pos': SITE_REV_POS, 'rev_neg': SITE_REV_NEG}[kind]
            width = shape[-1]
            nrows = int(np.prod(shape[:-1]))
            m = drop_mult(np.arange(nrows)[:, None], np.arange(width)[None, :], site, self.step, self.seed, self.p)
            return m.reshape(shape)
        return super().mult(shape, kind, call)

    def tok(self, shape, which):
        if self.corrupt_rate <= 0.0:
            return None
        site = SITE_TOK_POS if which == 'pos' else SITE_TOK_NEG
        N, WL = shape
        return torch.from_numpy(drop_mult(np.arange(N)[:, None], np.arange(WL)[None, :], site, self.step, self.seed,
                                          self.corrupt_rate))
