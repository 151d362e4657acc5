"""GRPO on the countdown task: make an equation over the given numbers that hits the target.

python examples/countdown/countdown.py --out /tmp/areal/data/countdown
python -m areal_vllm_amd.launcher.local examples/countdown/train.py --config examples/countdown/train_config.yaml
"""
import sys

from areal_vllm_amd.reward.countdown import countdown_reward_fn
from areal_vllm_amd.trainer.rl import run_grpo


def main(args):
    run_grpo(args, reward_fn=countdown_reward_fn)


if __name__ == "__main__":
    main(sys.argv[1:])
