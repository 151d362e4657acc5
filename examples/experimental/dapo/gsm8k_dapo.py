"""DAPO on GSM8K: the GRPO loop with the algorithm selected entirely by the YAML next to this file.

python -m areal_vllm_amd.launcher.local examples/experimental/dapo/gsm8k_dapo.py --config examples/experimental/dapo/gsm8k_dapo.yaml
"""
import sys

from areal_vllm_amd.reward.gsm8k import gsm8k_reward_fn
from areal_vllm_amd.trainer.rl import run_grpo


def main(args):
    run_grpo(args, reward_fn=gsm8k_reward_fn)


if __name__ == "__main__":
    main(sys.argv[1:])
