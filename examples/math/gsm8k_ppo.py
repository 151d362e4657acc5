"""PPO with a learned critic on GSM8K (actor + critic engines on the same ranks).

python -m areal_vllm_amd.launcher.local examples/math/gsm8k_ppo.py --config examples/math/gsm8k_ppo.yaml
"""
import sys

from areal_vllm_amd.api.cli_args import PPOConfig
from areal_vllm_amd.reward.gsm8k import gsm8k_reward_fn
from areal_vllm_amd.trainer.rl import run_grpo


def main(args):
    run_grpo(args, reward_fn=gsm8k_reward_fn, config_cls=PPOConfig)


if __name__ == "__main__":
    main(sys.argv[1:])
