"""ORACLE (test infrastructure only) - functional restatement of the reference's Ranger update,
``lib/torch_utils/solver/ranger.py:102-202`` (RAdam + Lookahead + gradient centralization), and of the train
loop's gradient clean-up ``core/catre/engine/engine.py:351-353``.  Pinned to the reference class itself through
``tests/golden/ranger_steps.npz`` (``oracle/make_golden.py``)."""
import math

import torch


def radam_terms(step, beta1, beta2, threshold=5):
    """(step_size, adaptive) of ranger.py:150-170."""
    beta2_t = beta2 ** step
    n_max = 2 / (1 - beta2) - 1
    n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma > threshold:
        size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2))
        return size / (1 - beta1 ** step), True
    return 1.0 / (1 - beta1 ** step), False


def ranger_step(p, grad, state, lr, betas=(0.95, 0.999), eps=1e-5, weight_decay=0.0, alpha=0.5, k=6, threshold=5,
                use_gc=True, gc_threshold=1, storage=None):
    """One update of one tensor; ``state`` is a dict that this function creates / advances in place.
    ``storage`` (e.g. torch.float32): round the parameter and the slow weights to that dtype where the reference, whose
    tensors ARE fp32, stores them (after the RAdam update, after the lookahead merge) - arithmetic stays in p's dtype."""
    def stored(t):
        return t if storage is None else t.to(storage).to(t.dtype)
    if not state:
        state.update(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p), slow_buffer=p.clone())
    beta1, beta2 = betas
    g = grad.clone()
    if use_gc and g.dim() > gc_threshold:
        g = g - g.mean(dim=tuple(range(1, g.dim())), keepdim=True)
    state["step"] += 1
    state["exp_avg_sq"] = state["exp_avg_sq"] * beta2 + (1 - beta2) * g * g
    state["exp_avg"] = state["exp_avg"] * beta1 + (1 - beta1) * g
    size, adaptive = radam_terms(state["step"], beta1, beta2, threshold)
    if weight_decay != 0:
        p = p - weight_decay * lr * p
    if adaptive:
        p = p - size * lr * state["exp_avg"] / (state["exp_avg_sq"].sqrt() + eps)
    else:
        p = p - size * lr * state["exp_avg"]
    p = stored(p)
    if state["step"] % k == 0:
        state["slow_buffer"] = stored(state["slow_buffer"] + alpha * (p - state["slow_buffer"]))
        p = state["slow_buffer"].clone()
    return p


def clean_grad(g, limit=1e5):
    return torch.nan_to_num(g, nan=0.0, posinf=limit, neginf=-limit)
