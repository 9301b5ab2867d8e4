"""Oracle (TEST INFRASTRUCTURE): clip-by-global-norm + dense Adam, restated.

Follows ``Optimizer.step`` (models/optimizers.py:205-243) as configured by
``build_optim`` (models/ps_model.py:20-35) and ``Optimizer.set_parameters``
(models/optimizers.py:165-187): ``_step += 1``; noam learning rate only when
``decay_method == 'noam'`` (:214-219, :233-237); ``clip_grad_norm_(params,
max_grad_norm)`` over every parameter that has a gradient (:241-242); then
``torch.optim.Adam(lr, betas, eps=1e-9, weight_decay=l2_lambda)`` (:186-187, :243).

The Adam arithmetic is the pinned third-party dependency (PyTorch, unpinned by
the reference; 2.10 here): ``m.lerp_(g, 1-b1)``; ``v = v*b2 + (1-b2) g*g``;
``p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)``.  Parameters whose
gradient is ``None`` are skipped by both the clip and Adam, as in the reference.
"""
import math
import torch


class ClipAdam(object):
    def __init__(self, lr, max_grad_norm=5.0, beta1=0.9, beta2=0.999, eps=1e-9,
                 weight_decay=0.0, decay_method='adam', warmup_steps=8000):
        self.learning_rate = lr
        self.original_lr = lr
        self.max_grad_norm = max_grad_norm
        self.betas = (beta1, beta2)
        self.eps = eps
        self.weight_decay = weight_decay
        self.decay_method = decay_method
        self.warmup_steps = warmup_steps
        self._step = 0
        self.state = {}       # name -> dict(step, m, v)
        self.last_total_norm = None

    def step(self, P, grads, touched=None):
        """In-place update of the tensors in ``P`` (name -> tensor) from ``grads``
        (name -> tensor or None).  Returns the pre-clip global grad norm.

        ``touched`` (name -> int64 row ids) selects the ROW-SPARSE rule for those tables (not the
        reference's: ``torch.optim.SparseAdam`` semantics with the global step in the bias correction):
        only the listed rows of p, m, v move; all other rows are left alone.  The clip norm is
        unchanged — untouched rows have a zero gradient."""
        touched = touched or {}
        self._step += 1
        if self.decay_method == 'noam':                      # optimizers.py:214-219
            self.learning_rate = self.original_lr * min(
                self._step ** (-0.5), self._step * self.warmup_steps ** (-1.5))
        names = [n for n in P if grads.get(n) is not None]
        # torch.nn.utils.clip_grad_norm_: L2 norm of the per-tensor L2 norms,
        # coef = max_norm / (total + 1e-6) clamped to 1, grads scaled in place.
        if self.max_grad_norm:
            norms = torch.stack([torch.linalg.vector_norm(grads[n], 2.0) for n in names])
            total = torch.linalg.vector_norm(norms, 2.0)
            coef = torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0)
            self.last_total_norm = float(total)
        else:
            coef = None
        b1, b2 = self.betas
        lr = self.learning_rate
        for n in names:
            g = grads[n] * coef if coef is not None else grads[n]
            p = P[n]
            st = self.state.setdefault(n, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
            st['step'] += 1
            t = st['step']
            bc1 = 1 - b1 ** t
            bc2 = 1 - b2 ** t
            step_size = lr / bc1
            if n in touched:
                rows = torch.as_tensor(touched[n], dtype=torch.int64)
                pr, gr, mr, vr = p[rows], g[rows], st['m'][rows], st['v'][rows]
                if self.weight_decay != 0:
                    gr = gr + self.weight_decay * pr
                mr.lerp_(gr, 1 - b1)
                vr.mul_(b2).addcmul_(gr, gr, value=1 - b2)
                pr.addcdiv_(mr, (vr.sqrt() / math.sqrt(bc2)).add_(self.eps), value=-step_size)
                p[rows], st['m'][rows], st['v'][rows] = pr, mr, vr
                continue
            if self.weight_decay != 0:
                g = g + self.weight_decay * p
            st['m'].lerp_(g, 1 - b1)
            st['v'].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (st['v'].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(st['m'], denom, value=-step_size)
        return self.last_total_norm
