"""Fused skinny-M GEMMs for the decode path (``csrc/hip/skinny_gemm.hip``).

``skinny_linear`` computes ``Y = A @ W.T`` for a handful of rows while streaming ``W`` exactly once,
with optional fusions that remove the elementwise kernels of a decoder layer:

* ``norm=(rowsumsq, weight, eps, gemma)``: the A operand is ``rmsnorm(x)`` applied on the fly; the
  per-row sum of squares comes from the previous GEMM's residual epilogue (or ``embed_rowsumsq``).
* ``residual=(R, rowsumsq_out)``: ``Y = bf16(bf16(acc) + R)`` is written as the new residual stream
  and its per-row sum of squares is accumulated into ``rowsumsq_out``.
* ``swiglu=True``: ``W`` is ``[gate; up]`` (2F rows); output is ``silu(gate) * up`` of shape ``[M, F]``.

The reference uses separate cuBLAS GEMMs plus elementwise kernels (SGLang / HF model code); on MI355X a
decode step is bound by weight streaming and per-kernel launch floor, which is what these fusions cut.
CPU tensors take a PyTorch path with the same rounding points.
"""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib as L

PRO_NONE, PRO_NORM = 0, 1
EPI_STORE, EPI_RESIDUAL, EPI_SWIGLU = 0, 1, 2


class SkinnyWorkspace:
    """fp32 split-K accumulation tile + ticket counters (both kept zero between calls by the kernel)."""

    def __init__(self, max_m: int, max_n: int, device):
        self.ws = torch.zeros(max_m * max_n, dtype=torch.float32, device=device)
        self.counters = torch.zeros(max(1, (max_m + 15) // 16) * max(1, max_n // 16), dtype=torch.int32, device=device)
        self.max_m, self.max_n = max_m, max_n


def pick_ksplit(M: int, N: int, K: int, nt: int, n_cu: int = 256) -> int:
    """Enough waves to cover the chip a few times over, each keeping a K range of at least 256."""
    waves = max(1, N // (16 * nt))
    want = 4 * n_cu
    s = max(1, min(want // waves, K // 256))
    return int(s)


def skinny_linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
                  norm: Optional[Tuple[torch.Tensor, torch.Tensor, float, bool]] = None,
                  residual: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, swiglu: bool = False,
                  out: Optional[torch.Tensor] = None, ksplit: Optional[int] = None, nt: Optional[int] = None,
                  workspace: Optional[SkinnyWorkspace] = None) -> torch.Tensor:
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty(M, n_out, dtype=x.dtype, device=x.device)
    if not L.use_hip(x):
        a = x
        if norm is not None:
            rss, nw, eps, gemma = norm
            rstd = torch.rsqrt(rss[:M].float() / K + eps).unsqueeze(-1)
            wf = nw.float() + (1.0 if gemma else 0.0)
            a = (x.float() * rstd * wf).to(x.dtype)
        y = F.linear(a, w, bias)
        if swiglu:
            g, u = y[:, :n_out].float(), y[:, n_out:].float()
            out.copy_((F.silu(g) * u).to(x.dtype))
        elif residual is not None:
            R, rss_out = residual
            o = (y.float() + R.float()).to(x.dtype)
            rss_out[:M] += o.float().pow(2).sum(-1)
            out.copy_(o)
        else:
            out.copy_(y)
        return out
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.stride(1) == 1 and w.is_contiguous()
    if nt is None:
        nt = 2 if (swiglu or N % 32 == 0 and N >= 8192) else 1
    if ksplit is None:
        ksplit = pick_ksplit(M, n_out if swiglu else N, K, 1 if swiglu else nt)
    if ksplit > 1:
        assert workspace is not None and workspace.max_m >= M and workspace.max_n >= N, "split-K needs a workspace"
    pro = PRO_NORM if norm is not None else PRO_NONE
    epi = EPI_SWIGLU if swiglu else (EPI_RESIDUAL if residual is not None else EPI_STORE)
    rss_in, nw, eps, gemma = norm if norm is not None else (None, None, 0.0, False)
    R, rss_out = residual if residual is not None else (None, None)
    L.check(L.lib().areal_skinny_gemm(
        L.ptr(x), L.c_int64(x.stride(0)), L.ptr(w), L.ptr(bias), L.ptr(out), L.c_int64(out.stride(0)), L.c_int(M),
        L.c_int(N), L.c_int(K), L.c_int(pro), L.ptr(rss_in), L.ptr(nw), L.c_float(eps), L.c_int(int(bool(gemma))),
        L.c_int(epi), L.ptr(R), L.c_int64(R.stride(0) if R is not None else 0), L.ptr(rss_out), L.c_int(ksplit),
        L.ptr(workspace.ws if workspace is not None else None),
        L.ptr(workspace.counters if workspace is not None else None), L.c_int(nt), L.stream_ptr()), "skinny_gemm")
    return out


def embed_rowsumsq(ids: torch.Tensor, table: torch.Tensor, out: torch.Tensor, rss0: torch.Tensor,
                   rss_clear: Optional[torch.Tensor] = None, scale: float = 1.0) -> torch.Tensor:
    """out[m] = table[ids[m]] * scale; rss0[m] = sum(out[m]^2); `rss_clear` (any shape) is zero-filled."""
    M, H = ids.numel(), table.shape[1]
    if not L.use_hip(table):
        v = F.embedding(ids, table)
        if scale != 1.0:
            v = (v.float() * scale).to(table.dtype)
        out[:M].copy_(v)
        if rss_clear is not None:
            rss_clear.zero_()
        rss0[:M] = v.float().pow(2).sum(-1)
        return out
    L.check(L.lib().areal_embed_rss(L.ptr(ids), L.ptr(table), L.ptr(out), L.ptr(rss0), L.ptr(rss_clear),
                                    L.c_int64(rss_clear.numel() if rss_clear is not None else 0), L.c_int(M),
                                    L.c_int(H), L.c_float(scale), L.stream_ptr()), "embed_rss")
    return out
