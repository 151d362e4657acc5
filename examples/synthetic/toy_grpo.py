"""Offline GRPO smoke run: a tiny random-init model learns to emit low token ids (dense synthetic reward).
Needs no tokenizer, dataset download or GPU; the rollout engine is colocated with the trainer.

python -m areal_vllm_amd.launcher.local examples/synthetic/toy_grpo.py \
    --config examples/synthetic/toy_grpo.yaml
"""
import sys

from areal_vllm_amd.trainer.rl import run_grpo


def main(args):
    run_grpo(args, load_tokenizer=False)


if __name__ == "__main__":
    main(sys.argv[1:])
