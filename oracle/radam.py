"""Oracle restatement of RAdam.step (ZEGGS/optimizers.py:31-99), numpy float32.

TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import math

import numpy as np


def radam_scalars(step, lr, beta1=0.9, beta2=0.999):
    """Host scalars of optimizers.py:64-84 for a given (1-based) step count.
    Returns (rectified: bool, step_scale) where the update is
      rectified : p -= step_scale * m / (sqrt(v) + eps)
      otherwise : p -= step_scale * m          (degenerated_to_sgd=True)"""
    beta2_t = beta2 ** step
    n_max = 2.0 / (1.0 - beta2) - 1.0
    n_sma = n_max - 2.0 * step * beta2_t / (1.0 - beta2_t)
    if n_sma >= 5:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma
                              * n_max / (n_max - 2)) / (1 - beta1 ** step)
        return True, step_size * lr
    return False, lr / (1 - beta1 ** step)


def radam_step(p, g, m, v, step, lr, eps, beta1=0.9, beta2=0.999, weight_decay=0.0):
    """In-place update of float32 arrays p, m, v with gradient g (step is 1-based).  weight_decay: optimizers.py:88-95
    (p += -weight_decay * lr * p before the update; degenerated_to_sgd=True applies a step in both branches)."""
    f = np.float32
    v *= f(beta2)
    v += f(1 - beta2) * g * g                       # addcmul_(grad, grad, value=1-beta2)
    m *= f(beta1)
    m += f(1 - beta1) * g                           # add_(grad, alpha=1-beta1)
    rect, scale = radam_scalars(step, lr, beta1, beta2)
    if weight_decay != 0:
        p += f(-weight_decay * lr) * p                # add_(p, alpha=-weight_decay * lr)
    if rect:
        p += f(-scale) * (m / (np.sqrt(v) + f(eps)))  # addcdiv_(m, sqrt(v)+eps, value=-step*lr)
    else:
        p += f(-scale) * m
    return p, m, v
