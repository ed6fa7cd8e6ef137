"""RAdam with the reference's constructor / state_dict surface, executed by one fused HIP kernel.

Mirrors `RAdam(params, lr, betas, eps, weight_decay, degenerated_to_sgd)` of the reference
(ZEGGS/optimizers.py:7-99).  When every parameter is a view of one flat fp32 buffer
(zeggs.engine.flatten_parameters) the whole step is a single launch over p/g/m/v; otherwise one
launch per tensor.  The host scalars (N_sma, step_size) follow optimizers.py:64-84 exactly.
"""
import math

import torch
from torch.optim.optimizer import Optimizer

from . import ops


def radam_scalars(step, lr, beta1, beta2, degenerated_to_sgd=True):
    """-> (rectified, step_scale, active)"""
    beta2_t = beta2 ** step
    n_max = 2 / (1 - beta2) - 1
    n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma >= 5:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max
                              / (n_max - 2)) / (1 - beta1 ** step)
        return True, step_size * lr, True
    if degenerated_to_sgd:
        return False, lr / (1 - beta1 ** step), True
    return False, 0.0, False


class RAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, degenerated_to_sgd=True):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameters: {}".format(betas))
        if not 0.0 <= weight_decay:
            raise ValueError("Invalid weight_decay value: {}".format(weight_decay))
        self.degenerated_to_sgd = degenerated_to_sgd
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                        buffer=[[None, None, None] for _ in range(10)])
        super().__init__(params, defaults)
        self._flat = None      # (p, g, m, v) flat buffers when attached
        self._step = 0
        self._guard = None     # (status words, all-reduced flag or None): the flat step is skipped ON THE DEVICE on a give-up
        self.early_pieces = 0
        self._early = []       # [lo, hi) slices of the flat buffers the coming step() has already been applied to (early())

    def attach_guard(self, status, gflag=None):
        """zeggs_radam_step_guarded for the flat step: `status` = the engine's sticky give-up words, `gflag` = the device
        float that carries every rank's flag through the gradient all-reduce (data-parallel runs)."""
        self._guard = (status, gflag)

    def rewind(self, nsteps):
        """Steps the device skipped (status[1]) never happened: the bias corrections must not count them."""
        self._step = max(0, self._step - int(nsteps))
        for group in self.param_groups:
            for q in group["params"]:
                if "step" in self.state[q]:
                    self.state[q]["step"] = self._step

    def attach_flat(self, flat_p, flat_g, keep_state=False):
        """Run the step as ONE kernel over flat buffers (all params must be views of flat_p in order).
        keep_state=True copies already-loaded per-tensor moments (resume) into the flat moment buffers."""
        self._flat = (flat_p, flat_g, torch.zeros_like(flat_p), torch.zeros_like(flat_p))
        off = 0
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state[p]
                n = p.numel()
                m_view, v_view = self._flat[2][off:off + n].view_as(p), self._flat[3][off:off + n].view_as(p)
                if keep_state and "exp_avg" in st:
                    m_view.copy_(st["exp_avg"])
                    v_view.copy_(st["exp_avg_sq"])
                    self._step = max(self._step, int(st.get("step", 0)))
                st["exp_avg"], st["exp_avg_sq"] = m_view, v_view
                st.setdefault("step", 0)
                off += n
        assert off == flat_p.numel()

    def _flat_piece(self, group, step, lo, hi, count):
        beta1, beta2 = group["betas"]
        rect, scale, active = radam_scalars(step, group["lr"], beta1, beta2, self.degenerated_to_sgd)
        p, g, m, v = (t[lo:hi] for t in self._flat)
        st, gf = self._guard if self._guard is not None else (None, None)
        ops.radam_step(p, g, m, v, beta1, beta2, group["eps"], scale if active else 0.0, rect, st, gf, count=count,
                       decay=group["weight_decay"] * group["lr"] if active else 0.0)

    @torch.no_grad()
    def early(self, lo, hi):
        """Flat mode: apply the COMING step() to elements [lo, hi) now, on the current stream -- their gradients are final there
        while others are still being computed (zeggs.engine: the decoder's slice, on the weight-gradient stream underneath the
        encoders' backward).  step() then covers the rest.  lo, hi multiples of 4 (16-byte accesses)."""
        assert self._flat is not None and len(self.param_groups) == 1 and lo % 4 == 0 and hi % 4 == 0 and lo < hi
        assert all(hi <= a or b <= lo for a, b in self._early), "early(): overlapping slices"
        self._flat_piece(self.param_groups[0], self._step + 1, lo, hi, count=False)
        self._early.append((lo, hi))
        self.early_pieces += 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._step += 1
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            if self._flat is not None:
                step = self._step
                n, at, rest = self._flat[0].numel(), 0, []
                for a, b in sorted(self._early):          # what early() has not done yet
                    if a > at:
                        rest.append((at, a))
                    at = b
                if at < n:
                    rest.append((at, n))
                self._early = []
                for k, (a, b) in enumerate(rest):         # (a skipped step is counted once: by the first piece here)
                    self._flat_piece(group, step, a, b, count=k == 0)
                for q in group["params"]:
                    self.state[q]["step"] = step
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                rect, scale, active = radam_scalars(st["step"], group["lr"], beta1, beta2, self.degenerated_to_sgd)
                pd, gd = p.data.view(-1), p.grad.data.contiguous().view(-1)
                ops.radam_step(pd, gd, st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1), beta1, beta2, group["eps"],
                               scale if active else 0.0, rect, decay=group["weight_decay"] * group["lr"] if active else 0.0)
        return loss
