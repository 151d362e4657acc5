"""Two decoder layers of Qwen2-1.5B (full vocabulary) through the train engine: forward log-probs and one
SFT-style train step on a packed 16k-token batch. Small enough to run under rocprofv3 --pmc; touches the
trainer-side kernels (flash fwd/bwd, RMSNorm fwd/bwd, SwiGLU fwd/bwd, RoPE, fused LM-head log-prob
fwd/bwd, grad norm, AdamW)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
os.environ.setdefault("AREAL_LOG_LEVEL", "WARNING")

from areal_vllm_amd.api.cli_args import MicroBatchSpec, OptimizerConfig, TrainEngineConfig  # noqa: E402
from areal_vllm_amd.api.io_struct import FinetuneSpec  # noqa: E402
from areal_vllm_amd.engine.train_engine import NativeTrainEngine  # noqa: E402
from areal_vllm_amd.models.config import PRESETS  # noqa: E402

PRESETS["qwen2-1.5b-2l"] = dict(PRESETS["qwen2-1.5b"], n_layers=2)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = TrainEngineConfig(experiment_name="p", trial_name="p", path="qwen2-1.5b-2l", init_from_scratch=True,
                        dtype="bfloat16", gradient_checkpointing=False, optimizer=OptimizerConfig(lr=1e-5),
                        mb_spec=MicroBatchSpec(max_tokens_per_mb=16384))
eng = NativeTrainEngine(cfg)
eng.create_process_group()
eng.initialize(None, FinetuneSpec(1, 64, 8))
g = torch.Generator().manual_seed(0)
lens = [1150] * 14
S = max(lens)
ids = torch.randint(2, 150000, (len(lens), S), generator=g)
am = torch.ones(len(lens), S, dtype=torch.bool)
lm = am.clone(); lm[:, :128] = False
batch = dict(input_ids=ids, attention_mask=am, loss_mask=lm)


def loss(proxy, mb):
    lp = proxy.logprobs(torch.roll(mb["input_ids"], -1, 0))
    m = mb["loss_mask"].float()
    return -(lp * m).sum() / m.sum()


for _ in range(steps):
    eng.forward(batch, post_hook=lambda p, mb: p.logprobs(torch.roll(mb["input_ids"], -1, 0)))
    st = eng.train_batch(batch, loss, lambda mb: mb["loss_mask"].float().sum())
torch.cuda.synchronize()
print("train probe ok", st)
