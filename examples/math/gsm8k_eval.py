"""Evaluate a checkpoint on the GSM8K test split through the rollout servers only
(allocation mode `native.d8p1t1+eval`; no trainer GPUs).

python -m areal_vllm_amd.launcher.local examples/math/gsm8k_eval.py --config examples/math/gsm8k_grpo.yaml \
    allocation_mode=native.d8p1t1+eval
"""
import os
import sys

from areal_vllm_amd.api.cli_args import GRPOConfig, load_expr_config
from areal_vllm_amd.dataset import get_custom_dataset
from areal_vllm_amd.engine.remote_engine import RemoteInferenceEngine
from areal_vllm_amd.reward.gsm8k import gsm8k_reward_fn
from areal_vllm_amd.utils import seeding, stats_tracker
from areal_vllm_amd.utils.dataloader import StatefulDataLoader
from areal_vllm_amd.utils.hf_utils import load_hf_tokenizer
from areal_vllm_amd.utils.stats_logger import StatsLogger
from areal_vllm_amd.workflow.rlvr import RLVRWorkflow


def main(args):
    config, _ = load_expr_config(args, GRPOConfig)
    rank = int(os.getenv("RANK", "0"))
    tokenizer = load_hf_tokenizer(config.tokenizer_path)
    seeding.set_random_seed(config.seed, f"eval{rank}")
    valid = get_custom_dataset(path=config.valid_dataset.path, rank=0, world_size=1, split="test",
                               type=config.valid_dataset.type, tokenizer=tokenizer)
    loader = StatefulDataLoader(valid, batch_size=config.valid_dataset.batch_size, shuffle=False, drop_last=False)
    config.rollout.max_head_offpolicyness = int(1e12)
    rollout = RemoteInferenceEngine(config.rollout)
    rollout.initialize()
    for t in (tokenizer.pad_token_id, tokenizer.eos_token_id):
        if t not in config.gconfig.stop_token_ids:
            config.gconfig.stop_token_ids.append(t)
    workflow = RLVRWorkflow(gsm8k_reward_fn, config.gconfig, tokenizer, rollout_stat_scope="eval-rollout",
                            dump_dir=os.path.join(StatsLogger.get_log_path(config.stats_logger), "generated-eval"))
    cnt = 0
    for data in loader:
        for item in data:
            rollout.submit(item, workflow)
            cnt += 1
    rollout.wait(cnt, timeout=None)
    print(f"Evaluation results: {stats_tracker.export_all()}")
    rollout.destroy()


if __name__ == "__main__":
    main(sys.argv[1:])
