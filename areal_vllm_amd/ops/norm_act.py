"""RMSNorm (fused residual add), SwiGLU and rotary embedding ops.

GPU tensors run the hand-written gfx950 kernels in ``csrc/hip/norm_act.hip``; CPU tensors run the
PyTorch reference below (same rounding points: the residual sum is rounded to bf16 before it is
normalised). Reference counterparts: ``realhf/impl/model/modules/mlp.py:319-341`` (RMSNorm),
``realhf/impl/model/modules/rotary.py`` (RoPE), HF Qwen2 MLP (SwiGLU).
"""
from typing import Optional, Tuple

import torch

from . import _lib as L


# ------------------------------------------------------------------------------- RMSNorm
def _rmsnorm_ref(x, res, w, eps, gemma):
    s = x if res is None else (x.float() + res.float()).to(x.dtype)
    sf = s.float()
    rstd = torch.rsqrt(sf.pow(2).mean(-1, keepdim=True) + eps)
    wf = w.float() + (1.0 if gemma else 0.0)
    y = (sf * rstd * wf).to(x.dtype)
    return y, s, rstd.squeeze(-1)


def _rmsnorm_bwd_ref(dy, x, w, rstd, dres, gemma):
    xf = x.float()
    xh = xf * rstd.unsqueeze(-1)
    wf = w.float() + (1.0 if gemma else 0.0)
    g = dy.float() * wf
    dot = (g * xh).mean(-1, keepdim=True)
    dx = rstd.unsqueeze(-1) * (g - xh * dot)
    if dres is not None:
        dx = dx + dres.float()
    dw = (dy.float() * xh).sum(0)
    return dx.to(x.dtype), dw.to(w.dtype)


class _RMSNormFn(torch.autograd.Function):
    """y, new_residual = rmsnorm(x + residual). `residual` may be None (then new_residual is x)."""

    @staticmethod
    def forward(ctx, x, residual, weight, eps, gemma):
        H = x.shape[-1]
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        r2 = None
        if residual is not None:
            r2 = residual.reshape(-1, H)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        T = x2.shape[0]
        if L.use_hip(x2):
            assert x2.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            y = torch.empty_like(x2)
            s = torch.empty_like(x2) if r2 is not None else x2
            rstd = torch.empty(T, dtype=torch.float32, device=x.device)
            L.check(
                L.lib().areal_rmsnorm_fwd(
                    L.ptr(x2), L.ptr(r2), L.ptr(weight), L.ptr(y), L.ptr(s if r2 is not None else None),
                    L.ptr(rstd), L.c_int(T), L.c_int(H), L.c_float(eps), L.c_int(int(gemma)), L.stream_ptr(),
                ),
                "rmsnorm_fwd",
            )
        else:
            y, s, rstd = _rmsnorm_ref(x2, r2, weight, eps, gemma)
        ctx.save_for_backward(s, weight, rstd)
        ctx.gemma = gemma
        ctx.has_res = residual is not None
        ctx.shape = x.shape
        return y.view(x.shape), s.view(x.shape)

    @staticmethod
    def backward(ctx, dy, ds):
        s, weight, rstd = ctx.saved_tensors
        H = s.shape[-1]
        T = s.shape[0]
        dy2 = dy.reshape(-1, H).contiguous()
        ds2 = ds.reshape(-1, H).contiguous() if ds is not None else None
        if L.use_hip(s):
            dx = torch.empty_like(s)
            dw = torch.empty_like(weight)
            G = L.lib().areal_rmsnorm_bwd_nblocks(L.c_int(T))
            ws = torch.empty(G * H, dtype=torch.float32, device=s.device)
            L.check(
                L.lib().areal_rmsnorm_bwd(
                    L.ptr(dy2), L.ptr(s), L.ptr(weight), L.ptr(rstd), L.ptr(ds2), L.ptr(dx), L.ptr(dw),
                    L.ptr(None), L.ptr(ws), L.c_int(T), L.c_int(H), L.c_int(int(ctx.gemma)), L.stream_ptr(),
                ),
                "rmsnorm_bwd",
            )
        else:
            dx, dw = _rmsnorm_bwd_ref(dy2, s, weight, rstd, ds2, ctx.gemma)
        dx = dx.view(ctx.shape)
        return dx, (dx if ctx.has_res else None), dw, None, None


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, residual: Optional[torch.Tensor] = None,
            gemma: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (normed, new_residual) where new_residual = x + residual (or x)."""
    return _RMSNormFn.apply(x, residual, weight, eps, gemma)


# ------------------------------------------------------------------- q / k norm inside a packed qkv
def _qk_norm_ref(qkv3, qw, kw, eps, hq, hkv):
    x = qkv3.float()
    n = hq + hkv
    rstd = torch.rsqrt(x[:, :n].pow(2).mean(-1, keepdim=True) + eps)
    w = torch.cat([qw.float().expand(hq, -1), kw.float().expand(hkv, -1)], 0)
    y = torch.cat([x[:, :n] * rstd * w, x[:, n:]], 1)
    return y.to(qkv3.dtype), rstd.squeeze(-1)


class _QKNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, qw, kw, eps, hq, hkv, D):
        T = qkv.shape[0]
        x = qkv.reshape(T, hq + 2 * hkv, D)
        if not x.is_contiguous():
            x = x.contiguous()
        hip = L.use_hip(x) and x.dtype == torch.bfloat16 and D in (64, 128, 256)
        # the result is allocated in the caller's shape (RoPE rotates it in place afterwards: it must not be a view
        # made inside this function)
        out = torch.empty(qkv.shape, dtype=x.dtype, device=x.device)
        if hip:
            rstd = torch.empty(T, hq + hkv, dtype=torch.float32, device=x.device)
            L.check(L.lib().areal_qk_norm_fwd(L.ptr(x), L.ptr(qw.contiguous()), L.ptr(kw.contiguous()), L.ptr(out),
                                              L.ptr(rstd), L.c_int64(T), L.c_int(hq), L.c_int(hkv), L.c_int(D),
                                              L.c_float(eps), L.stream_ptr()), "qk_norm_fwd")
        else:
            y, rstd = _qk_norm_ref(x, qw, kw, eps, hq, hkv)
            out.view(T, hq + 2 * hkv, D).copy_(y)
        ctx.save_for_backward(x, qw, kw, rstd)
        ctx.dims, ctx.hip, ctx.shape = (hq, hkv, D), hip, qkv.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        x, qw, kw, rstd = ctx.saved_tensors
        hq, hkv, D = ctx.dims
        T = x.shape[0]
        dy = dout.reshape(T, hq + 2 * hkv, D).contiguous()
        if ctx.hip:
            lib = L.lib()
            dx = torch.empty_like(x)
            dw = torch.empty(2, D, dtype=x.dtype, device=x.device)
            G = int(lib.areal_qk_norm_bwd_nblocks(L.c_int64(T * (hq + 2 * hkv))))
            ws = torch.empty(G * 2 * D, dtype=torch.float32, device=x.device)
            L.check(lib.areal_qk_norm_bwd(L.ptr(dy), L.ptr(x), L.ptr(qw.contiguous()), L.ptr(kw.contiguous()),
                                          L.ptr(rstd), L.ptr(dx), L.ptr(dw), L.ptr(ws), L.c_int64(T), L.c_int(hq),
                                          L.c_int(hkv), L.c_int(D), L.stream_ptr()), "qk_norm_bwd")
            dqw, dkw = dw[0].to(qw.dtype), dw[1].to(kw.dtype)
        else:
            n = hq + hkv
            xf, g0 = x.float(), dy.float()
            w = torch.cat([qw.float().expand(hq, -1), kw.float().expand(hkv, -1)], 0)
            xh = xf[:, :n] * rstd.unsqueeze(-1)
            g = g0[:, :n] * w
            dxn = rstd.unsqueeze(-1) * (g - xh * (g * xh).mean(-1, keepdim=True))
            dx = torch.cat([dxn, g0[:, n:]], 1).to(x.dtype).contiguous()
            dwf = g0[:, :n] * xh
            dqw, dkw = dwf[:, :hq].sum((0, 1)).to(qw.dtype), dwf[:, hq:].sum((0, 1)).to(kw.dtype)
        return dx.reshape(ctx.shape), dqw, dkw, None, None, None, None


def qk_norm(qkv: torch.Tensor, q_weight: torch.Tensor, k_weight: torch.Tensor, eps: float, hq: int, hkv: int,
            D: int) -> torch.Tensor:
    """Per-head RMSNorm of the q and k heads of a packed projection output ``qkv`` [T, (hq + 2 hkv) * D] (Qwen3's
    q_norm / k_norm), the v heads copied: one pass forward and one backward (``csrc/hip/norm_act.hip``) instead of
    slicing the heads out, normalising and concatenating. Same shape out."""
    return _QKNormFn.apply(qkv, q_weight, k_weight, eps, hq, hkv, D)


# -------------------------------------------------------------------------------- SwiGLU
GLU_ACTS = {"silu": 0, "gelu_tanh": 1, "gelu_pytorch_tanh": 1}


def _act_ref(g, act: int):
    return torch.nn.functional.silu(g) if act == 0 else torch.nn.functional.gelu(g, approximate="tanh")


def _swiglu_ref(gu, act: int = 0):
    F = gu.shape[-1] // 2
    g, u = gu[..., :F].float(), gu[..., F:].float()
    return (_act_ref(g, act) * u).to(gu.dtype)


class _SwiGLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gu, act):
        F = gu.shape[-1] // 2
        gu2 = gu.reshape(-1, 2 * F)
        if not gu2.is_contiguous():
            gu2 = gu2.contiguous()
        if L.use_hip(gu2):
            assert gu2.dtype == torch.bfloat16
            out = torch.empty(gu2.shape[0], F, dtype=gu.dtype, device=gu.device)
            L.check(L.lib().areal_glu_fwd(L.ptr(gu2), L.ptr(out), L.c_int64(gu2.shape[0]), L.c_int(F), L.c_int(act),
                                          L.stream_ptr()), "glu_fwd")
        else:
            out = _swiglu_ref(gu2, act)
        ctx.save_for_backward(gu2)
        ctx.shape, ctx.act = gu.shape, act
        return out.view(*gu.shape[:-1], F)

    @staticmethod
    def backward(ctx, dy):
        (gu2,) = ctx.saved_tensors
        F = gu2.shape[-1] // 2
        dy2 = dy.reshape(-1, F).contiguous()
        if L.use_hip(gu2):
            dgu = torch.empty_like(gu2)
            L.check(L.lib().areal_glu_bwd(L.ptr(dy2), L.ptr(gu2), L.ptr(dgu), L.c_int64(gu2.shape[0]),
                                          L.c_int(F), L.c_int(ctx.act), L.stream_ptr()), "glu_bwd")
        else:
            with torch.enable_grad():
                g = gu2[:, :F].float().detach().requires_grad_(True)
                a = _act_ref(g, ctx.act)
                (da,) = torch.autograd.grad(a.sum(), g)
            u, d = gu2[:, F:].float(), dy2.float()
            dgu = torch.cat([d * u * da, d * a.detach()], -1).to(gu2.dtype)
        return dgu.view(ctx.shape), None


def swiglu(gate_up: torch.Tensor, act: str = "silu") -> torch.Tensor:
    """out = act(gate_up[..., :F]) * gate_up[..., F:] for the fused gate|up projection; `act` is "silu"
    (SwiGLU) or "gelu_tanh" (GeGLU, Gemma)."""
    return _SwiGLUFn.apply(gate_up, GLU_ACTS[act])


# ---------------------------------------------------------------------------------- RoPE
class RotaryTable:
    """Host-precomputed fp32 cos/sin table [max_pos, D/2] (half-split layout)."""

    def __init__(self, head_dim: int, max_pos: int, base: float = 10000.0, device="cpu", scaling: float = 1.0):
        self.head_dim = head_dim
        self.max_pos = max_pos
        inv = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float64) / head_dim))
        t = torch.arange(max_pos, dtype=torch.float64) / scaling
        fr = torch.outer(t, inv)
        self.cos = fr.cos().float().to(device).contiguous()
        self.sin = fr.sin().float().to(device).contiguous()

    def to(self, device):
        self.cos = self.cos.to(device)
        self.sin = self.sin.to(device)
        return self


def _rope_ref(x, pos, table: RotaryTable, n_rot, sign):
    D = x.shape[-1]
    half = D // 2
    p = pos.long().clamp(0, table.max_pos - 1)
    c = table.cos[p].unsqueeze(1)
    s = table.sin[p].unsqueeze(1) * sign
    xr = x[:, :n_rot].float()
    lo, hi = xr[..., :half], xr[..., half:]
    out = torch.cat([lo * c - hi * s, hi * c + lo * s], -1).to(x.dtype)
    x[:, :n_rot] = out
    return x


def rope_inplace_(x: torch.Tensor, positions: torch.Tensor, table: RotaryTable, n_rot_heads: int, sign: float = 1.0):
    """Rotate the first `n_rot_heads` heads of x [T, H, D] in place. sign=-1 gives the inverse."""
    assert x.dim() == 3 and x.is_contiguous()
    T, H, D = x.shape
    if L.use_hip(x):
        assert x.dtype == torch.bfloat16 and positions.dtype == torch.int32
        L.check(
            L.lib().areal_rope_inplace(
                L.ptr(x), L.ptr(positions), L.ptr(table.cos), L.ptr(table.sin), L.c_int64(T), L.c_int(H),
                L.c_int(n_rot_heads), L.c_int(D), L.c_float(sign), L.c_int(table.max_pos), L.stream_ptr(),
            ),
            "rope_inplace",
        )
        return x
    return _rope_ref(x, positions, table, n_rot_heads, sign)


class _RopeFn(torch.autograd.Function):
    """In-place RoPE on the 2-D output of the fused qkv GEMM (viewed as [T, heads, D] internally)."""

    @staticmethod
    def forward(ctx, qkv2d, positions, table, n_rot_heads, head_dim):
        ctx.table = table
        ctx.n_rot = n_rot_heads
        ctx.head_dim = head_dim
        ctx.save_for_backward(positions)
        ctx.mark_dirty(qkv2d)
        rope_inplace_(qkv2d.view(qkv2d.shape[0], -1, head_dim), positions, table, n_rot_heads, 1.0)
        return qkv2d

    @staticmethod
    def backward(ctx, d):
        (positions,) = ctx.saved_tensors
        # The incoming gradient is the freshly allocated dqkv of the attention backward and has
        # no other consumer, so the inverse rotation is applied in place.
        d = d.contiguous()
        rope_inplace_(d.view(d.shape[0], -1, ctx.head_dim), positions, ctx.table, ctx.n_rot, -1.0)
        return d, None, None, None, None


def apply_rope(qkv2d: torch.Tensor, positions: torch.Tensor, table: RotaryTable, n_rot_heads: int,
               head_dim: int) -> torch.Tensor:
    """Differentiable in-place RoPE on the q and k heads of a fused [T, (Hq+2Hkv)*D] projection."""
    return _RopeFn.apply(qkv2d, positions, table, n_rot_heads, head_dim)
