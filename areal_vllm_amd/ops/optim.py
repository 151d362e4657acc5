"""Flat-buffer grad-norm / scale / fused AdamW ops.

Counterparts: ``areal/utils/fsdp.py:27-60,116-234`` (multi_tensor_l2norm / scale, clip) and
``torch.optim.AdamW`` at ``areal/engine/base_hf_engine.py:197-246``. The optimizer state vector lives
on device so that a step (norm -> clip coefficient -> skip-if-nonfinite -> AdamW) never synchronises
with the host: [sumsq, norm, clip_coef, applied, step, bias_corr1, bias_corr2, pad].
"""
import math

import torch

from . import _lib as L

STATE_SIZE = 8


def sumsq_(x: torch.Tensor, out: torch.Tensor) -> None:
    """out[0] += sum(x**2) in fp32 (x: flat bf16 or fp32)."""
    if L.use_hip(x):
        L.check(L.lib().areal_sumsq(L.ptr(x), L.c_int(L.dtype_flag(x)), L.c_int64(x.numel()), L.ptr(out),
                                    L.stream_ptr()), "sumsq")
    else:
        out[0] += x.float().pow(2).sum()


def scale_(x: torch.Tensor, scale: float, scale_tensor: torch.Tensor = None) -> None:
    if L.use_hip(x):
        L.check(L.lib().areal_scale(L.ptr(x), L.c_int(L.dtype_flag(x)), L.c_int64(x.numel()), L.ptr(scale_tensor),
                                    L.c_float(scale), L.stream_ptr()), "scale")
    else:
        s = scale * (float(scale_tensor[0]) if scale_tensor is not None else 1.0)
        x.mul_(s)


def adam_prepare_(state: torch.Tensor, max_norm: float, grad_prescale: float, beta1: float, beta2: float) -> None:
    """state[0] holds the (already all-reduced) sum of squares of the unscaled gradient."""
    if L.use_hip(state):
        L.check(L.lib().areal_adam_prepare(L.ptr(state), L.c_float(max_norm), L.c_float(grad_prescale),
                                           L.c_float(beta1), L.c_float(beta2), L.stream_ptr()), "adam_prepare")
        return
    norm = math.sqrt(max(float(state[0]), 0.0)) * abs(grad_prescale) if math.isfinite(float(state[0])) else float("nan")
    ok = math.isfinite(norm)
    coef = grad_prescale
    if max_norm > 0 and ok:
        coef *= min(1.0, max_norm / (norm + 1e-6))
    state[1] = norm
    state[2] = coef
    state[3] = 1.0 if ok else 0.0
    if ok:
        state[4] += 1.0
    step = max(float(state[4]), 1.0)
    state[5] = 1.0 - beta1 ** step
    state[6] = 1.0 - beta2 ** step


def adamw_(master, m, v, grad, shadow, state, lr, beta1, beta2, eps, weight_decay) -> None:
    """One AdamW update on a flat shard (fp32 master/m/v; grad bf16 or fp32; shadow bf16 or None)."""
    n = master.numel()
    if L.use_hip(master):
        L.check(
            L.lib().areal_adamw(L.ptr(master), L.ptr(m), L.ptr(v), L.ptr(grad), L.c_int(L.dtype_flag(grad)),
                                L.ptr(shadow), L.c_int64(n), L.ptr(state), L.c_float(lr), L.c_float(beta1),
                                L.c_float(beta2), L.c_float(eps), L.c_float(weight_decay), L.stream_ptr()),
            "adamw",
        )
        return
    if float(state[3]) == 0.0:
        return
    g = grad.float() * float(state[2])
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = float(state[5]), float(state[6])
    denom = v.sqrt() / math.sqrt(bc2) + eps
    master.mul_(1 - lr * weight_decay).addcdiv_(m, denom, value=-lr / bc1)
    if shadow is not None:
        shadow.copy_(master)
