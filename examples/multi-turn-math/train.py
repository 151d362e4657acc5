"""Multi-turn math: the model may retry after a fixed feedback message; reward is discounted per turn.

python -m areal_vllm_amd.launcher.local examples/multi-turn-math/train.py --config examples/multi-turn-math/config.yaml
"""
import os
import sys
from dataclasses import dataclass

from areal_vllm_amd.api.cli_args import GRPOConfig
from areal_vllm_amd.reward.gsm8k import gsm8k_reward_fn
from areal_vllm_amd.trainer.rl import run_grpo
from areal_vllm_amd.workflow.multi_turn import MultiTurnWorkflow


@dataclass
class MultiTurnGRPOConfig(GRPOConfig):
    max_turns: int = 2
    turn_discount: float = 0.9


def build(config, tokenizer, reward_fn, log_dir):
    mk = lambda g, scope, sub: MultiTurnWorkflow(  # noqa: E731
        reward_fn, g, tokenizer, config.max_turns, config.turn_discount, rollout_stat_scope=scope,
        dump_dir=os.path.join(log_dir, sub))
    return (mk(config.gconfig, "rollout", "generated"),
            mk(config.gconfig.new(temperature=0.6), "eval-rollout", "generated-eval"))


def main(args):
    run_grpo(args, reward_fn=gsm8k_reward_fn, workflow_builder=build, config_cls=MultiTurnGRPOConfig)


if __name__ == "__main__":
    main(sys.argv[1:])
