"""Time the varlen flash-attention forward / backward kernels on a trainer-shaped packed batch.

    python tools/attn_probe.py                      # 25 sequences of ~1150 tokens, 12 q / 2 kv heads (Qwen2-1.5B)
    python tools/attn_probe.py 24576 28 4           # one sequence of 24576 tokens, 28 q / 4 kv heads (Qwen2.5-7B)
    python tools/attn_probe.py 1100,1100,... 14 2 64  # head size 64 (Qwen2-0.5B)
"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from areal_vllm_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
Hq, Hkv, D = 12, 2, 128
lens = torch.randint(1050, 1216, (25,)).tolist()
if len(sys.argv) > 1:
    lens = [int(x) for x in sys.argv[1].split(",")]
    Hq, Hkv = int(sys.argv[2]), int(sys.argv[3])
    if len(sys.argv) > 4:
        D = int(sys.argv[4])  # head sizes below 128 run zero-padded on the MFMA kernels
T = sum(lens)
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
qkv = torch.randn(T, Hq + 2 * Hkv, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
def fwd():
    return ops.flash_attn_varlen_qkv(qkv, cu, max(lens), Hq, Hkv)
out = fwd(); g = torch.randn_like(out)
def fb():
    qkv.grad = None
    fwd().backward(g)
def bench(fn, n=20 if sum(lens) < 60000 else 5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
with torch.no_grad():
    tf = bench(fwd)
tfb = bench(fb)
fl = sum(4 * l * l / 2 * Hq * D for l in lens)
print(f"T={T} fwd {tf*1e6:.0f}us ({fl/tf/1e12:.0f} TF/s)  fwd+bwd {tfb*1e6:.0f}us  bwd~{(tfb-tf)*1e6:.0f}us ({2.5*fl/(tfb-tf)/1e12:.0f} TF/s)")
if D != 128:  # what the per-sequence library attention would cost for the same batch
    from areal_vllm_amd.ops import attention as A
    def fb_sdpa():
        qkv.grad = None
        q, k, v = qkv.split([Hq, Hkv, Hkv], 1)
        A.attention_varlen_sdpa(q, k, v, cu, cu, D ** -0.5, True).backward(g)
    print(f"per-sequence library attention fwd+bwd {bench(fb_sdpa)*1e6:.0f}us")
