"""GRPO on GSM8K with verifiable rewards (single turn).

python -m areal_vllm_amd.launcher.local examples/math/gsm8k_grpo.py --config examples/math/gsm8k_grpo.yaml
"""
import sys

from areal_vllm_amd.reward.gsm8k import gsm8k_reward_fn
from areal_vllm_amd.trainer.rl import run_grpo


def main(args):
    run_grpo(args, reward_fn=gsm8k_reward_fn)


if __name__ == "__main__":
    main(sys.argv[1:])
