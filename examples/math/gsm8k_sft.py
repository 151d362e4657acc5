"""Supervised fine-tuning on GSM8K question/answer pairs.

python -m areal_vllm_amd.launcher.local examples/math/gsm8k_sft.py --config examples/math/gsm8k_sft.yaml
"""
import sys

from areal_vllm_amd.trainer.sft import run_sft


def main(args):
    run_sft(args)


if __name__ == "__main__":
    main(sys.argv[1:])
