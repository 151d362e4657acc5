"""GRPO on GSM8K with a TP x PP trainer (the reference's gsm8k_grpo_megatron.py).

python -m areal_vllm_amd.launcher.local examples/experimental/gsm8k_grpo_tp_pp.py --config examples/experimental/configs/gsm8k_grpo_tp_pp.yaml
"""
import sys

from areal_vllm_amd.reward.gsm8k import gsm8k_reward_fn
from areal_vllm_amd.trainer.rl import run_grpo


def main(args):
    run_grpo(args, reward_fn=gsm8k_reward_fn)


if __name__ == "__main__":
    main(sys.argv[1:])
