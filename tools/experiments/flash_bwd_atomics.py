"""Experiment (not part of the library): how much of the flash-attention backward time is the fp32
atomic accumulation of dQ? Builds a scratch copy of csrc/hip/flash_attn.hip in /tmp whose dQ atomics are
replaced by a never-taken store (the dQ MFMAs stay), compiles it with hipcc and times both variants on the
trainer's packed-batch shape. The scratch variant computes WRONG dQ by construction; nothing here is
loaded by the package."""
import ctypes
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from areal_vllm_amd.ops import _lib as L  # noqa: E402
from areal_vllm_amd.ops import attention as A  # noqa: E402

src = open(os.path.join(ROOT, "csrc/hip/flash_attn.hip")).read()
needle = "atomicAdd(dq_acc + ((size_t)(q_beg + qrow) * Hq + head) * D + 32 * wid + r,\n                    dqacc[reg] * softmax_scale);"
assert needle in src
variant = src.replace(needle, "{ if (dqacc[reg] == 12345.678f) dq_acc[0] = 1.f; }")
os.makedirs("/tmp/flash_exp", exist_ok=True)
open("/tmp/flash_exp/flash_noatomic.hip", "w").write(variant)
so = "/tmp/flash_exp/libflash_noatomic.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-I", os.path.join(ROOT, "csrc/hip"),
                       "-o", so, "/tmp/flash_exp/flash_noatomic.hip"])
alt = ctypes.CDLL(so)

dev = torch.device("cuda")
torch.manual_seed(0)
Hq, Hkv, D = 12, 2, 128
lens = torch.randint(1050, 1216, (25,)).tolist()
T = sum(lens)
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
qkv = torch.randn(T, Hq + 2 * Hkv, D, device=dev, dtype=torch.bfloat16)
q, k, v = qkv[:, :Hq], qkv[:, Hq:Hq + Hkv], qkv[:, Hq + Hkv:]
out, lse = A._flash_fwd_hip(q, k, v, cu, cu, max(lens), D ** -0.5, True)
dout = torch.randn_like(out)
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[:, :Hq], dqkv[:, Hq:Hq + Hkv], dqkv[:, Hq + Hkv:]
delta = torch.empty(Hq, T, dtype=torch.float32, device=dev)
dq_acc = torch.zeros(T, Hq, D, dtype=torch.float32, device=dev)


def call(lib):
    s = A._strides
    rc = lib.areal_flash_bwd(
        L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dq_acc), L.ptr(dq),
        L.ptr(dk), L.ptr(dv), L.ptr(cu), L.ptr(cu), L.c_int(len(lens)), L.c_int(max(lens)), L.c_int(T), L.c_int(Hq),
        L.c_int(Hkv), L.c_int(D), L.c_int64(s(q)), L.c_int64(s(k)), L.c_int64(s(v)), L.c_int64(s(out)),
        L.c_int64(s(dout)), L.c_int64(s(dq)), L.c_int64(s(dk)), L.c_int64(s(dv)), L.c_float(D ** -0.5), L.c_int(1),
        L.stream_ptr())
    assert rc == 0, rc


def bench(lib, n=20):
    for _ in range(3):
        call(lib)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        call(lib)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


fl = 2.5 * sum(4 * l * l / 2 * Hq * D for l in lens)
a, b = bench(L.lib()), bench(alt)
print(f"T={T}: backward with dQ atomics {a:.0f} us ({fl / a / 1e6:.0f} TF/s), without {b:.0f} us ({fl / b / 1e6:.0f} TF/s)")
