"""GAE and interval gather/scatter ops (HIP on GPU, PyTorch reference on CPU).

Semantics follow the reference's native extensions: ``csrc/cugae/gae.cu`` (called from
``realhf/impl/model/utils/ppo_functional.py:326-367``) and ``csrc/interval_op/*`` (called from
``realhf/impl/model/nn/flatten_param.py:97-186``); the padded variant implements the AReaL-lite
actor's loop (``areal/engine/ppo/actor.py:131-147``) in one launch.
"""
from typing import List, Tuple

import numpy as np
import torch

from . import _lib as L


# ----------------------------------------------------------------------------------- GAE
def gae_1d_packed_ref(rewards, values, cu_seqlens, bootstrap, gamma, lam):
    adv = torch.zeros_like(rewards)
    ret = torch.zeros_like(rewards)
    cu = cu_seqlens.tolist()
    for i in range(len(cu) - 1):
        rs, re = cu[i], cu[i + 1]
        vs = rs + i
        last = 0.0
        for t in reversed(range(re - rs)):
            nv = values[vs + t + 1] if (t != re - rs - 1 or bool(bootstrap[i])) else 0.0
            delta = rewards[rs + t] + gamma * nv - values[vs + t]
            last = delta + gamma * lam * last
            adv[rs + t] = last
            ret[rs + t] = last + values[vs + t]
    return adv, ret


def gae_1d_packed(rewards: torch.Tensor, values: torch.Tensor, cu_seqlens: torch.Tensor, bootstrap: torch.Tensor,
                  gamma: float, lam: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """rewards [total], values [total + bs] (one bootstrap slot per sequence), cu_seqlens int32 [bs+1],
    bootstrap bool [bs] (True: the final step bootstraps from its extra value slot)."""
    bs = cu_seqlens.numel() - 1
    assert values.numel() == rewards.numel() + bs
    if not L.use_hip(rewards):
        return gae_1d_packed_ref(rewards, values, cu_seqlens, bootstrap, gamma, lam)
    rewards = rewards.float().contiguous()
    values = values.float().contiguous()
    adv = torch.zeros_like(rewards)
    ret = torch.zeros_like(rewards)
    boot = bootstrap.to(torch.uint8).contiguous()
    L.check(
        L.lib().areal_gae_1d_packed(L.ptr(rewards), L.ptr(values), L.ptr(cu_seqlens.int().contiguous()), L.ptr(boot),
                                    L.ptr(adv), L.ptr(ret), L.c_int(bs), L.c_float(gamma), L.c_float(lam),
                                    L.stream_ptr()),
        "gae_1d_packed",
    )
    return adv, ret


def gae_padded_ref(rewards, values, no_eos, gamma, lam):
    B, S = rewards.shape
    adv = torch.zeros_like(rewards)
    last = torch.zeros(B, dtype=rewards.dtype, device=rewards.device)
    for t in reversed(range(S - 1)):
        nv = values[:, t + 1]
        if t == S - 2:
            nv = nv * no_eos.to(values.dtype)
        delta = rewards[:, t] + gamma * nv - values[:, t]
        last = delta + gamma * lam * last
        adv[:, t] = last
    return adv, adv + values


def gae_padded(rewards: torch.Tensor, values: torch.Tensor, no_eos: torch.Tensor, gamma: float, lam: float):
    """Padded [B, S] GAE; column S-1 of the advantage is 0; the last step bootstraps only where no_eos."""
    if not L.use_hip(rewards):
        return gae_padded_ref(rewards.float(), values.float(), no_eos, gamma, lam)
    B, S = rewards.shape
    rewards = rewards.float().contiguous()
    values = values.float().contiguous()
    adv = torch.empty_like(rewards)
    ret = torch.empty_like(rewards)
    ne = no_eos.to(torch.uint8).contiguous()
    L.check(
        L.lib().areal_gae_padded(L.ptr(rewards), L.ptr(values), L.ptr(ne), L.ptr(adv), L.ptr(ret), L.c_int(B),
                                 L.c_int(S), L.c_float(gamma), L.c_float(lam), L.stream_ptr()),
        "gae_padded",
    )
    return adv, ret


def gae_2d_ref(rewards, values, done, done_idx, cu_dones, flag, mode, gamma, lam):
    B, H = rewards.shape
    adv = torch.zeros_like(rewards)
    ret = torch.zeros_like(rewards)
    c = gamma * lam
    cu = cu_dones.tolist()
    for b in range(B):
        d0, nd = cu[b], cu[b + 1] - cu[b]
        for j in range(nd + 1):
            is_last = j == nd
            t_start = 0 if j == 0 else int(done_idx[d0 + j - 1])
            t_end = H if is_last else int(done_idx[d0 + j])
            if mode == 0:
                last_nv = 0.0 if (t_end > t_start and bool(done[b, t_end])) else float(values[b, t_end])
                if (not is_last) and t_end > t_start and bool(flag[d0 + j]):
                    t_end -= 1
                    adv[b, t_end] = 0.0
                    ret[b, t_end] = values[b, t_end]
                    last_nv = float(values[b, t_end])
            else:
                if not is_last:
                    t_end -= 1
                    last_nv = float(values[b, t_end]) if (t_end > t_start and bool(flag[d0 + j])) else 0.0
                else:
                    last_nv = float(values[b, H])
            last = 0.0
            for t in reversed(range(t_start, t_end)):
                nv = last_nv if t == t_end - 1 else float(values[b, t + 1])
                delta = float(rewards[b, t]) + gamma * nv - float(values[b, t])
                last = delta + c * last
                adv[b, t] = last
                ret[b, t] = last + values[b, t]
    return adv, ret


def gae_2d(rewards, values, done, done_idx, cu_dones, flag, max_num_dones: int, gamma: float, lam: float,
           overlapped: bool = True):
    """Padded GAE with episode boundaries. rewards [B, H]; values, done [B, H+1]; done_idx int32 [n_done]
    (time index of every boundary, grouped per row by cu_dones int32 [B+1]); flag bool [n_done] is
    `is_truncate` for the overlapped form and `bootstrap` for the non-overlapped form (see rl_ops.hip)."""
    mode = 0 if overlapped else 1
    if not L.use_hip(rewards):
        return gae_2d_ref(rewards.float(), values.float(), done, done_idx, cu_dones, flag, mode, gamma, lam)
    B, H = rewards.shape
    rewards = rewards.float().contiguous()
    values = values.float().contiguous()
    adv = torch.zeros_like(rewards)
    ret = torch.zeros_like(rewards)
    L.check(
        L.lib().areal_gae_2d(L.ptr(rewards), L.ptr(values), L.ptr(done.to(torch.uint8).contiguous()),
                             L.ptr(done_idx.int().contiguous()), L.ptr(cu_dones.int().contiguous()),
                             L.ptr(flag.to(torch.uint8).contiguous()), L.ptr(adv), L.ptr(ret), L.c_int(B), L.c_int(H),
                             L.c_int(int(max_num_dones)), L.c_int(mode), L.c_float(gamma), L.c_float(lam),
                             L.stream_ptr()),
        "gae_2d",
    )
    return adv, ret


# ------------------------------------------------------------------------------ interval ops
def merge_intervals(intervals: np.ndarray) -> np.ndarray:
    """Sort [n, 2] (start, end) intervals and merge the ones that touch or overlap (host; the C++
    runtime version is used when built, reference: csrc/interval_op/interval_op.cpp:1-29)."""
    try:
        from areal_vllm_amd._C import _runtime

        return np.asarray(_runtime.merge_intervals(np.ascontiguousarray(intervals, dtype=np.int64))).reshape(-1, 2)
    except ImportError:
        pass
    if len(intervals) == 0:
        return np.zeros((0, 2), dtype=np.int64)
    iv = intervals[np.argsort(intervals[:, 0], kind="stable")]
    out = [list(iv[0])]
    for s, e in iv[1:]:
        if s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return np.asarray(out, dtype=np.int64)


def _interval_meta(intervals, device):
    iv = torch.as_tensor(np.ascontiguousarray(intervals, dtype=np.int64))
    lens = iv[:, 1] - iv[:, 0]
    offs = torch.cumsum(lens, 0) - lens
    return iv.to(device), offs.to(device), int(lens.sum()), int(lens.max()) if len(lens) else 0


def slice_intervals(flat: torch.Tensor, intervals: np.ndarray) -> torch.Tensor:
    """Concatenate flat[s:e] for every interval into one packed tensor."""
    iv, offs, total, max_len = _interval_meta(intervals, flat.device)
    out = torch.empty(total, dtype=flat.dtype, device=flat.device)
    if iv.shape[0] == 0:
        return out
    if L.use_hip(flat):
        L.check(
            L.lib().areal_interval_copy(L.ptr(flat), L.ptr(out), L.ptr(iv), L.ptr(offs), L.c_int(iv.shape[0]),
                                        L.c_int64(max_len), L.c_int(flat.element_size()), L.c_int(0), L.stream_ptr()),
            "interval_copy",
        )
        return out
    return torch.cat([flat[int(s):int(e)] for s, e in intervals])


def set_intervals(src: torch.Tensor, flat: torch.Tensor, intervals: np.ndarray) -> None:
    """Scatter the packed `src` back into flat[s:e] for every interval."""
    iv, offs, total, max_len = _interval_meta(intervals, flat.device)
    assert src.numel() == total and src.dtype == flat.dtype
    if iv.shape[0] == 0:
        return
    if L.use_hip(flat):
        L.check(
            L.lib().areal_interval_copy(L.ptr(flat), L.ptr(src.contiguous()), L.ptr(iv), L.ptr(offs),
                                        L.c_int(iv.shape[0]), L.c_int64(max_len), L.c_int(flat.element_size()),
                                        L.c_int(1), L.stream_ptr()),
            "interval_copy",
        )
        return
    o = 0
    for s, e in intervals:
        n = int(e) - int(s)
        flat[int(s):int(e)] = src[o:o + n]
        o += n
