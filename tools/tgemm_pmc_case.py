"""One GEMM shape, library and hand-written kernel, a few calls each: the workload of a counter pass
(`bash tools/pmc_run.sh <tag> tools/tgemm_pmc_case.py [N K [M]]`)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from areal_vllm_amd.ops import train_gemm as tg  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 17920
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
M = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
dev = torch.device("cuda")
torch.manual_seed(0)
a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
b = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(4):
    torch.mm(a, b.t(), out=out)
    tg.tgemm_tn(a, b, out=out)
torch.cuda.synchronize()
