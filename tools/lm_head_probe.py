"""Per-rank cost of the LM head + log-prob loss with the vocabulary replicated vs sharded over tp.

    python tools/lm_head_probe.py --model qwen2.5-32b --tp 4 --tokens 16384

One GPU plays one tensor-parallel rank: it runs the fused LM head + loss (forward and backward, kept bf16
logits) once on the full [V, H] head - what every rank of a replicated head pays - and once on rank 0's
vocabulary shard. The cross-rank exchange of the sharded path is one all-gather of 16 bytes per token plus
the all-reduce of d(hidden) [T, H]; their sizes are printed, not timed (a single device has no peer).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from areal_vllm_amd import ops  # noqa: E402
from areal_vllm_amd.models.config import preset  # noqa: E402


def run(T, H, rows, vocab_start, vocab, chunk, iters, dev):
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    g = torch.Generator(device=dev).manual_seed(0)
    w = (torch.randn(rows, H, device=dev, generator=g) * 0.02).bfloat16().requires_grad_(True)
    h = torch.randn(T, H, device=dev, generator=g).bfloat16().requires_grad_(True)
    labels = torch.randint(0, vocab, (T,), device=dev, generator=g)
    times = []
    for it in range(iters + 2):
        w.grad = h.grad = None
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        lp, ent = ops.lm_head_logprobs(h, w, labels, 1.0, chunk, True, vocab_start=vocab_start)
        e1.record()
        (lp.sum() + 0.01 * ent.sum()).backward()
        e2.record()
        torch.cuda.synchronize()
        if it >= 2:
            times.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
    fwd = sum(t[0] for t in times) / len(times)
    bwd = sum(t[1] for t in times) / len(times)
    peak = torch.cuda.max_memory_allocated() - base
    return fwd, bwd, peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen2.5-32b")
    ap.add_argument("--tp", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=16384)
    ap.add_argument("--chunk", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = preset(args.model)
    V, H, T = cfg.vocab_size, cfg.hidden_size, args.tokens
    lo, hi = ops.vocab_shard_range(V, args.tp, 0)
    print(f"{args.model}: vocab {V}, hidden {H}, {T} tokens per micro-batch, tp {args.tp} -> shard rows {hi - lo}")
    for name, rows in (("replicated head (every tp rank)", V), (f"vocab shard (1 of {args.tp})", hi - lo)):
        fwd, bwd, peak = run(T, H, rows, 0, V, args.chunk, args.iters, dev)
        wbytes = rows * H * 2
        print(f"  {name:34s} fwd {fwd:7.2f} ms  bwd {bwd:7.2f} ms  total {fwd + bwd:7.2f} ms   "
              f"weight {wbytes / 2**20:7.1f} MiB, kept logits {T * rows * 2 / 2**20:7.1f} MiB, "
              f"peak above baseline {peak / 2**20:8.1f} MiB", flush=True)
    print(f"  sharded path exchanges per micro-batch: stats all-gather {T * 16 * args.tp / 2**10:.0f} KiB received, "
          f"d(hidden) all-reduce {T * H * 2 / 2**20:.1f} MiB")


if __name__ == "__main__":
    main()
