"""Hand-written dense bf16 GEMM for the trainer and for prefill (``csrc/hip/train_gemm.hip``).

``tgemm_tn(a, b)`` = ``a @ b.T`` for K-contiguous operands (activations x ``nn.Linear`` weight, or output gradient
x transposed weight copy). The kernel covers the shapes transformer layers produce (K a multiple of 128, N of 4,
16-byte aligned rows); `supported()` tells the caller when to keep the library GEMM instead - it is a statement
about shapes, not a fallback for a missing extension (on a GPU the extension is mandatory)."""
import os
from typing import Optional

import torch

from . import _lib as L

# AREAL_OWN_WGRAD=0 sends weight gradients back to the library GEMM (A/B runs on one box)
OWN_WGRAD = os.environ.get("AREAL_OWN_WGRAD", "1") != "0"


def supported(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None) -> bool:
    if not (a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2):
        return False
    M, K = a.shape
    N = b.shape[0]
    if b.shape[1] != K or K % 128 != 0 or N % 4 != 0 or M == 0:
        return False
    if a.stride(1) != 1 or b.stride(1) != 1 or a.stride(0) % 8 or b.stride(0) % 8:
        return False
    if a.data_ptr() % 16 or b.data_ptr() % 16:
        return False
    if bias is not None and (bias.dtype != torch.bfloat16 or not bias.is_contiguous() or bias.data_ptr() % 8):
        return False
    return True


def tgemm_tn(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None, stagger: bool = True, max_blocks: int = 256) -> torch.Tensor:
    """out[M, N] = a[M, K] @ b[N, K]^T (+ bias[N]); bf16 in, fp32 accumulation, bf16 out."""
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    assert supported(a, b, bias) and out.dtype == torch.bfloat16 and out.stride(1) == 1 and out.stride(0) % 4 == 0
    L.lib()
    L.check(L.lib().areal_tgemm_tn(L.ptr(a), L.c_int64(a.stride(0)), L.ptr(b), L.c_int64(b.stride(0)), L.ptr(out),
                                   L.c_int64(out.stride(0)), L.ptr(bias), L.c_int(M), L.c_int(N), L.c_int(K),
                                   L.c_int(int(stagger)), L.c_int(int(max_blocks)), L.stream_ptr()), "tgemm_tn")
    return out


def wgrad_supported(dy: torch.Tensor, x: torch.Tensor, acc: torch.Tensor) -> bool:
    if not OWN_WGRAD:
        return False
    if not (dy.is_cuda and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and acc.dtype == torch.float32):
        return False
    if dy.dim() != 2 or x.dim() != 2 or dy.shape[0] != x.shape[0] or acc.shape != (dy.shape[1], x.shape[1]):
        return False
    M, N = dy.shape
    K = x.shape[1]
    if N % 128 or K % 256 or M < 256:
        return False
    if dy.stride(1) != 1 or x.stride(1) != 1 or acc.stride(1) != 1 or dy.stride(0) % 8 or x.stride(0) % 8:
        return False
    return dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0


def tgemm_wgrad_(acc: torch.Tensor, dy: torch.Tensor, x: torch.Tensor, splits: int = 0, stagger: bool = True) -> torch.Tensor:
    """acc[N, K] (fp32) += dy[M, N]^T @ x[M, K]: the weight gradient of ``y = x @ W^T`` summed straight into the
    fp32 gradient buffer. The kernel reduces whole pairs of 64-token tiles; the last M % 128 tokens are added by a
    library GEMM on those few rows."""
    assert wgrad_supported(dy, x, acc)
    M = dy.shape[0]
    main = M - M % 128
    L.check(L.lib().areal_tgemm_wgrad(L.ptr(dy), L.c_int64(dy.stride(0)), L.ptr(x), L.c_int64(x.stride(0)), L.ptr(acc),
                                      L.c_int64(acc.stride(0)), L.c_int(main), L.c_int(dy.shape[1]), L.c_int(x.shape[1]),
                                      L.c_int(int(splits)), L.c_int(int(stagger)), L.stream_ptr()), "tgemm_wgrad")
    if main < M:
        torch.addmm(acc, dy[main:].t(), x[main:], out_dtype=torch.float32, out=acc)
    return acc


# rows from which the fused gate|up + SwiGLU kernel replaces library GEMM + activation kernel in no-grad forwards
GATED_MIN_ROWS = int(os.environ.get("AREAL_FUSED_MLP_MIN_ROWS", "4096"))


def gated_supported(x: torch.Tensor, w_gate_up: torch.Tensor) -> bool:
    if not (x.is_cuda and x.dim() == 2 and w_gate_up.dim() == 2 and x.dtype == torch.bfloat16
            and w_gate_up.dtype == torch.bfloat16):
        return False
    M, K = x.shape
    if w_gate_up.shape[1] != K or w_gate_up.shape[0] % 2 or K % 128 or (w_gate_up.shape[0] // 2) % 4:
        return False
    if x.stride(1) != 1 or x.stride(0) % 8 or not w_gate_up.is_contiguous():
        return False
    return x.data_ptr() % 16 == 0 and w_gate_up.data_ptr() % 16 == 0 and M > 0


def tgemm_swiglu(x: torch.Tensor, w_gate_up: torch.Tensor, out: Optional[torch.Tensor] = None,
                 max_blocks: int = 256) -> torch.Tensor:
    """out[M, F] = silu(x @ Wg^T) * (x @ Wu^T) for W = [Wg; Wu] of shape [2 F, K]: the gate|up projection of a gated
    MLP and its activation in one kernel (gate and up are rounded to bf16 before the activation, as the unfused path
    holds them); the [M, 2 F] intermediate is never written. For forward passes that keep no activations."""
    assert gated_supported(x, w_gate_up)
    M, K = x.shape
    Fo = w_gate_up.shape[0] // 2
    if out is None:
        out = torch.empty(M, Fo, dtype=torch.bfloat16, device=x.device)
    assert out.dtype == torch.bfloat16 and out.stride(1) == 1 and out.stride(0) % 4 == 0
    L.check(L.lib().areal_tgemm_swiglu(L.ptr(x), L.c_int64(x.stride(0)), L.ptr(w_gate_up), L.c_int64(w_gate_up.stride(0)),
                                       L.ptr(out), L.c_int64(out.stride(0)), L.c_int(M), L.c_int(Fo), L.c_int(K),
                                       L.c_int(int(max_blocks)), L.stream_ptr()), "tgemm_swiglu")
    return out


def gate_up_act(x: torch.Tensor, w_gate_up: torch.Tensor, act: str = "silu") -> Optional[torch.Tensor]:
    """The fused kernel's result when it applies (no gradient wanted, SiLU gate, enough rows, supported shapes),
    else None: the caller then runs projection and activation separately."""
    if act != "silu" or torch.is_grad_enabled() and (x.requires_grad or w_gate_up.requires_grad):
        return None
    x2 = x.reshape(-1, x.shape[-1])
    if x2.shape[0] < GATED_MIN_ROWS or not gated_supported(x2, w_gate_up):
        return None
    return tgemm_swiglu(x2, w_gate_up).view(*x.shape[:-1], w_gate_up.shape[0] // 2)
