"""Prefill throughput of the rollout engine, scheduler included: N prompts of L tokens each, one new token per
prompt (so a request is one prefill pass - several chunks for long prompts - plus the sampling of its first token).

    python tools/prefill_probe.py --model qwen2-1.5b --cases 256x512,64x2048,8x16384,1x24576

Prints tokens/s per case and the model FLOP rate it corresponds to (2 x non-embedding parameters per token plus
attention). Two rounds per case: the second one finds the decode / prefill kernels warm.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from areal_vllm_amd.api.cli_args import InferenceServerConfig  # noqa: E402
from areal_vllm_amd.inference.engine import NativeInferenceEngine  # noqa: E402
from areal_vllm_amd.models.config import preset  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen2-1.5b")
    ap.add_argument("--cases", default="256x512,64x2048,8x16384,1x24576")
    ap.add_argument("--max-prefill-tokens", type=int, default=32768)
    ap.add_argument("--kv-cache-gb", type=float, default=24.0)
    a = ap.parse_args()
    mc = preset(a.model)
    cases = [tuple(int(x) for x in c.split("x")) for c in a.cases.split(",")]
    ctx = max(l for _, l in cases) + 64
    srv = InferenceServerConfig(model_path=a.model, random_seed=1, dtype="bfloat16", context_length=ctx,
                                max_running_requests=max(n for n, _ in cases), max_prefill_tokens=a.max_prefill_tokens,
                                schedule_policy="fcfs")
    eng = NativeInferenceEngine(srv, model_config=mc, device=torch.device("cuda", 0), kv_cache_gb=a.kv_cache_gb)
    eng.initialize(load_weights=False)
    g = torch.Generator().manual_seed(0)
    body = 2.0 * mc.n_params(with_embedding=False) if hasattr(mc, "n_params") else None
    try:
        for n, l in cases:
            for rnd in range(2):
                prompts = [torch.randint(2, min(100000, mc.vocab_size), (l,), generator=g).tolist() for _ in range(n)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.generate_blocking(prompts, max_new_tokens=1, temperature=1.0)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            toks = n * l
            # attention: 4 x head_dim x q_heads x (context / 2) FLOP per token and layer (causal)
            attn = 4.0 * mc.head_dim * mc.n_q_heads * (l / 2.0) * mc.n_layers
            per_tok = (body or 0.0) + attn
            rate = f"{toks * per_tok / dt / 1e12:7.1f} TFLOP/s" if body else ""
            print(f"{n:4d} prompts x {l:6d} tokens: {dt * 1e3:8.1f} ms  {toks / dt:10.0f} tokens/s  {rate}", flush=True)
    finally:
        eng.destroy()


if __name__ == "__main__":
    main()
