"""Time the captured decode graph of the rollout engine at fixed batch / context sizes.

    python tools/decode_probe.py --model qwen2-1.5b --batch 128,256,384 --ctx 640 --iters 20

Builds the model with random weights, fabricates decode plans (sequential page tables, every sequence at
`ctx` tokens) and replays the `decode_steps`-step graph. Prints ms per decode step and tokens/s. Run it
under ``rocprofv3 --kernel-trace --stats`` for the per-kernel split (``--iters 3`` is enough then).

Two ways of driving the same plans are timed: plan by plan (upload everything, replay, wait for the tokens -
the device idles while the host books a plan and prepares the next) and pipelined (the continuation of a plan
is launched before the plan's tokens are awaited: only page tables travel, the device goes from one replay
into the next). The difference between the two is the host gap per plan.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from areal_vllm_amd.inference.model_runner import ModelRunner  # noqa: E402
from areal_vllm_amd.models.config import preset  # noqa: E402
from areal_vllm_amd.models.transformer import Transformer  # noqa: E402
from areal_vllm_amd.ops.attention import PAGE_SIZE  # noqa: E402


def make_plan(B, ctx, steps, max_pages, V, page_base=0, pages_per_seq=None):
    pages_per_seq = pages_per_seq or (ctx + steps + PAGE_SIZE - 1) // PAGE_SIZE + 1
    bt = np.zeros((B, pages_per_seq), dtype=np.int32)
    for b in range(B):
        bt[b] = page_base + b * pages_per_seq + np.arange(pages_per_seq)
    pos = np.full(B, ctx - 1, dtype=np.int32)
    slots = (bt[np.arange(B), pos // PAGE_SIZE].astype(np.int64) * PAGE_SIZE + pos % PAGE_SIZE)
    return dict(req_ids=list(range(B)), n_steps=steps, input_ids=np.random.randint(2, V, B).astype(np.int32),
                positions=pos, context_lens=np.full(B, ctx, dtype=np.int32), slot_mapping=slots, block_tables=bt,
                max_pages=pages_per_seq, temperature=np.ones(B, dtype=np.float32), top_p=np.ones(B, dtype=np.float32),
                top_k=np.full(B, 1 << 30, dtype=np.int64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen2-1.5b")
    ap.add_argument("--batch", default="128,256,384")
    ap.add_argument("--ctx", type=int, default=640)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--kv-dtype", default="auto", choices=["auto", "fp8_e4m3"], help="element type of the KV cache")
    ap.add_argument("--target-wgs", type=int, default=0, help="override the split-KV workgroup target (experiment)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="also time every batch split over this many runners (own KV cache, graphs and stream each, "
                         "shared weights) replaying concurrently: do the dependent-kernel gaps of one lane hide "
                         "behind the other lane's kernels?")
    args = ap.parse_args()
    if args.target_wgs:
        from areal_vllm_amd.ops import attention as _att

        _att.DECODE_TARGET_WORKGROUPS = _att.DECODE_TARGET_WORKGROUPS_FP8 = args.target_wgs
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = preset(args.model)
    model = Transformer(cfg, dtype=torch.bfloat16, device=dev)
    model.requires_grad_(False)
    model.eval()
    model.flatten()
    model.init_weights(seed=1)
    batches = [int(b) for b in args.batch.split(",")]
    maxb = max(batches)
    grow = args.steps * (args.iters + 2)  # every plan appends `steps` tokens to every sequence
    pages_per_seq = (args.ctx + grow + PAGE_SIZE - 1) // PAGE_SIZE + 1
    runner = ModelRunner(model, maxb * pages_per_seq + 8, args.ctx + grow + 64, maxb, decode_steps=args.steps,
                         use_graph=not args.no_graph, buckets=batches,
                         kv_dtype=torch.float8_e4m3fn if args.kv_dtype == "fp8_e4m3" else None)
    for kc, vc in runner.kv:  # non-trivial cache contents (softmax over zeros is cheaper than real data)
        if kc.dtype == torch.float8_e4m3fn:
            kc.view(torch.uint8).random_(0, 0x78)  # finite positive e4m3 codes
            vc.view(torch.uint8).random_(0, 0x40)
        else:
            kc.normal_(0, 0.5)
            vc.normal_(0, 0.5)
    for B in batches:
        plans = [make_plan(B, args.ctx + i * args.steps, args.steps, runner.max_pages, cfg.vocab_size,
                           pages_per_seq=pages_per_seq) for i in range(args.iters + 1)]
        runner.decode(plans[0])
        torch.cuda.synchronize()
        # plan by plan: full upload, replay, wait
        t0 = time.perf_counter()
        for i in range(args.iters):
            runner.decode(plans[i])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (args.iters * args.steps)
        print(f"B={B:4d} ctx={args.ctx}: {dt * 1e3:7.3f} ms/step  {B / dt:9.0f} tok/s  "
              f"({dt * 1e6 / cfg.n_layers:6.1f} us/layer incl. head+sampler)  [plan by plan]", flush=True)
        # pipelined: the continuation is launched before the previous plan's tokens are read
        cont = dict(req_ids=plans[0]["req_ids"], n_steps=args.steps, continuation=True, max_pages=pages_per_seq,
                    block_tables=plans[0]["block_tables"], kill=np.zeros(B, np.int32))
        runner.decode(plans[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cur = runner.decode_launch(plans[0])
        for i in range(1, args.iters):
            nxt = runner.decode_launch(cont)
            runner.decode_wait(cur)
            cur = nxt
        runner.decode_wait(cur)
        torch.cuda.synchronize()
        dp = (time.perf_counter() - t0) / (args.iters * args.steps)
        print(f"B={B:4d} ctx={args.ctx}: {dp * 1e3:7.3f} ms/step  {B / dp:9.0f} tok/s  "
              f"({dp * 1e6 / cfg.n_layers:6.1f} us/layer incl. head+sampler)  [pipelined]  "
              f"host gap removed: {(dt - dp) * args.steps * 1e3:6.3f} ms per {args.steps}-step plan", flush=True)
    if args.lanes > 1:
        del runner
        torch.cuda.empty_cache()
        probe_lanes(args, model, cfg, batches, pages_per_seq, grow)


def probe_lanes(args, model, cfg, batches, pages_per_seq, grow):
    dev = next(model.parameters()).device
    L = args.lanes
    for B in batches:
        Bh = B // L
        runners = [ModelRunner(model, Bh * pages_per_seq + 8, args.ctx + grow + 64, Bh, decode_steps=args.steps,
                               use_graph=True, buckets=[Bh]) for _ in range(L)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(L)]
        for r in runners:
            for kc, vc in r.kv:
                kc.normal_(0, 0.5)
                vc.normal_(0, 0.5)
        plan0 = make_plan(Bh, args.ctx, args.steps, runners[0].max_pages, cfg.vocab_size, pages_per_seq=pages_per_seq)
        cont = dict(req_ids=plan0["req_ids"], n_steps=args.steps, continuation=True, max_pages=pages_per_seq,
                    block_tables=plan0["block_tables"], kill=np.zeros(Bh, np.int32))
        for r, st in zip(runners, streams):
            with torch.cuda.stream(st):
                r.decode(plan0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cur = []
        for r, st in zip(runners, streams):
            with torch.cuda.stream(st):
                cur.append(r.decode_launch(plan0))
        for i in range(1, args.iters):
            for k, (r, st) in enumerate(zip(runners, streams)):
                with torch.cuda.stream(st):
                    nxt = r.decode_launch(cont)
                    r.decode_wait(cur[k])
                    cur[k] = nxt
        for k, (r, st) in enumerate(zip(runners, streams)):
            with torch.cuda.stream(st):
                r.decode_wait(cur[k])
        torch.cuda.synchronize()
        dp = (time.perf_counter() - t0) / (args.iters * args.steps)
        print(f"B={B:4d} as {L} lanes of {Bh} ctx={args.ctx}: {dp * 1e3:7.3f} ms per step of all lanes  "
              f"{Bh * L / dp:9.0f} tok/s  [pipelined, concurrent lanes]", flush=True)
        del runners
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
