"""Trainer-only throughput: SFT steps of a full model on one GPU (no rollout engine on the device).

    python tools/sft_bench.py --model qwen2-1.5b --seqs 128 --len 1150 --steps 4 --max-tokens-per-mb 32768

Prints tokens/s of forward-only scoring and of full train steps (forward + backward + ZeRO-1 AdamW) and the
model FLOP utilisation against 2.5 PFLOP/s dense bf16."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29562")
os.environ.setdefault("AREAL_LOG_LEVEL", "WARNING")

from areal_vllm_amd.api.cli_args import MicroBatchSpec, OptimizerConfig, TrainEngineConfig  # noqa: E402
from areal_vllm_amd.api.io_struct import FinetuneSpec  # noqa: E402
from areal_vllm_amd.engine.train_engine import NativeTrainEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="qwen2-1.5b")
ap.add_argument("--seqs", type=int, default=128)
ap.add_argument("--len", type=int, default=1150)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--max-tokens-per-mb", type=int, default=32768)
ap.add_argument("--gradient-checkpointing", action="store_true")
a = ap.parse_args()
cfg = TrainEngineConfig(experiment_name="p", trial_name="p", path=a.model, init_from_scratch=True, dtype="bfloat16",
                        gradient_checkpointing=a.gradient_checkpointing, optimizer=OptimizerConfig(lr=1e-5),
                        mb_spec=MicroBatchSpec(max_tokens_per_mb=a.max_tokens_per_mb))
eng = NativeTrainEngine(cfg)
eng.create_process_group()
eng.initialize(None, FinetuneSpec(1, 1024, a.seqs))
mc = eng.model_config
g = torch.Generator().manual_seed(0)
ids = torch.randint(2, min(150000, mc.vocab_size), (a.seqs, a.len), generator=g)
am = torch.ones(a.seqs, a.len, dtype=torch.bool)
batch = dict(input_ids=ids, attention_mask=am, loss_mask=am.clone())
score = lambda p, mb: p.logprobs(torch.roll(mb["input_ids"], -1, 0))  # noqa: E731


def loss(proxy, mb):
    m = mb["loss_mask"].float()
    return -(score(proxy, mb) * m).sum() / m.sum()


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / a.steps


T = a.seqs * a.len
n_params = mc.n_active_params()  # the embedding table is also the LM head GEMM for tied models; MoE: active experts
attn = 2 * 2 * a.len * mc.n_q_heads * mc.head_dim * mc.n_layers / 2  # causal QK^T + PV flops per token (forward)
fwd_flops = T * (2 * n_params + attn)
t_fwd = timed(lambda: eng.forward(batch, post_hook=score))
t_train = timed(lambda: eng.train_batch(batch, loss, lambda mb: mb["loss_mask"].float().sum()))
print(f"{a.model}: {T} tokens/step ({a.seqs} x {a.len}), micro-batches of <= {a.max_tokens_per_mb} tokens")
print(f"forward scoring : {t_fwd * 1e3:8.1f} ms/step  {T / t_fwd:9.0f} tokens/s  {fwd_flops / t_fwd / 1e12:6.0f} TFLOP/s "
      f"({100 * fwd_flops / t_fwd / 2.5e15:4.1f}% of dense bf16 peak)")
print(f"train step      : {t_train * 1e3:8.1f} ms/step  {T / t_train:9.0f} tokens/s  {3 * fwd_flops / t_train / 1e12:6.0f} TFLOP/s "
      f"({100 * 3 * fwd_flops / t_train / 2.5e15:4.1f}% of dense bf16 peak)")
print(f"peak device memory: {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB allocated, "
      f"{torch.cuda.max_memory_reserved() / 2**30:.1f} GiB reserved")
