"""Evaluate a checkpoint on the GSM8K test split through the rollout servers only
(allocation mode `native.d8p1t1+eval`; no trainer GPUs).

python -m areal_vllm_amd.launcher.local examples/math/gsm8k_eval.py --config examples/math/gsm8k_grpo.yaml \
    allocation_mode=native.d8p1t1+eval
"""
import json
import os
import sys

from areal_vllm_amd.api.cli_args import GRPOConfig, load_expr_config
from areal_vllm_amd.dataset import get_custom_dataset
from areal_vllm_amd.engine.remote_engine import RemoteInferenceEngine
from areal_vllm_amd.reward.gsm8k import gsm8k_reward_fn
from areal_vllm_amd.utils import seeding, stats_tracker
from areal_vllm_amd.utils.hf_utils import load_hf_tokenizer
from areal_vllm_amd.utils.stats_logger import StatsLogger
from areal_vllm_amd.workflow.rlvr import RLVRWorkflow

NO_STALENESS_LIMIT = 10 ** 12  # evaluation never waits for a trainer version


def build_workflow(config, tokenizer) -> RLVRWorkflow:
    stops = set(config.gconfig.stop_token_ids)
    stops.update(t for t in (tokenizer.pad_token_id, tokenizer.eos_token_id) if t is not None)
    config.gconfig.stop_token_ids = sorted(stops)
    dump_dir = os.path.join(StatsLogger.get_log_path(config.stats_logger), "generated-eval")
    return RLVRWorkflow(gsm8k_reward_fn, config.gconfig, tokenizer, rollout_stat_scope="eval-rollout",
                        dump_dir=dump_dir)


def evaluate(config) -> dict:
    tokenizer = load_hf_tokenizer(config.tokenizer_path)
    seeding.set_random_seed(config.seed, "eval" + os.environ.get("RANK", "0"))
    problems = get_custom_dataset(path=config.valid_dataset.path, rank=0, world_size=1, split="test",
                                  type=config.valid_dataset.type, tokenizer=tokenizer)
    config.rollout.max_head_offpolicyness = NO_STALENESS_LIMIT
    servers = RemoteInferenceEngine(config.rollout)
    servers.initialize()
    try:
        workflow = build_workflow(config, tokenizer)
        n = 0
        for n, problem in enumerate(problems, start=1):  # everything is queued at once: the servers batch it
            servers.submit(problem, workflow)
        servers.wait(n, timeout=None)
        return dict(n_problems=n, **stats_tracker.export_all())
    finally:
        servers.destroy()


if __name__ == "__main__":
    cfg, _ = load_expr_config(sys.argv[1:], GRPOConfig)
    print("Evaluation results: " + json.dumps(evaluate(cfg), default=float, sort_keys=True))
