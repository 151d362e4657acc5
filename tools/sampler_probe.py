"""Time the sampler on a decode batch of full-vocabulary logits: one workgroup per row against rows split over
several workgroups (``csrc/hip/sampling.hip``), replayed from a captured graph.

    python tools/sampler_probe.py --vocab 151936 --batch 128,256,384,512
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from areal_vllm_amd import ops  # noqa: E402
from areal_vllm_amd.ops import sampling  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(4):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=151936)
    ap.add_argument("--batch", default="64,128,256,320,384,512")
    ap.add_argument("--top-p", type=float, default=1.0, help="< 1: time the filtered path (nucleus sampling) instead")
    ap.add_argument("--top-k", type=int, default=0, help="> 0: time the filtered path (top-k) instead")
    ap.add_argument("--scale", type=float, default=2.0, help="standard deviation of the random logits")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    for B in [int(x) for x in a.batch.split(",")]:
        # two sets of logits (2 x B x V x 2 bytes): more than the 256 MB Infinity Cache from B = 512 on
        logits = [(a.scale * torch.randn(B, a.vocab, device=dev)).bfloat16() for _ in range(2)]
        temp = torch.ones(B, device=dev)
        step = torch.zeros(1, dtype=torch.int64, device=dev)
        tok = torch.empty(B, dtype=torch.int64, device=dev)
        lp = torch.empty(B, dtype=torch.float32, device=dev)
        i = [0]

        def run():
            i[0] ^= 1
            ops.sample(logits[i[0]], temp, None, None, seed=1, step=step, out_tokens=tok, out_logprobs=lp)

        if a.top_p < 1.0 or a.top_k > 0:
            tp = torch.full((B,), a.top_p, device=dev)
            tk = torch.full((B,), a.top_k if a.top_k > 0 else a.vocab, dtype=torch.int32, device=dev)

            def run_f():
                i[0] ^= 1
                ops.sample(logits[i[0]], temp, tk, tp, seed=1, step=step, out_tokens=tok, out_logprobs=lp)

            t = timeit(run_f)
            print(f"B={B:4d} V={a.vocab}: filtered (top_p={a.top_p}, top_k={a.top_k or 'off'}) {t:8.1f} us", flush=True)
            continue
        res = {}
        for split in (False, True):
            sampling.SPLIT_UNFILTERED = split
            res[split] = timeit(run)
        sampling.SPLIT_UNFILTERED = True
        mb = B * a.vocab * 2 / 1e6
        print(f"B={B:4d} V={a.vocab}: one workgroup per row {res[False]:7.1f} us   split rows {res[True]:7.1f} us "
              f"({mb / res[True] / 1e6 * 1e6 / 1e3:.2f} TB/s)   logits {mb:.0f} MB", flush=True)


if __name__ == "__main__":
    main()
