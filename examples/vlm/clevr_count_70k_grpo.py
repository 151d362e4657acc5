"""GRPO on CLEVR-Count-70k with a vision-language policy.

The workflow, dataset and reward are wired exactly as for text models; the policy must be a
vision-language checkpoint whose family the native model zoo implements.
"""
import os
import sys

from areal_vllm_amd.reward.clevr_count_70k import clevr_count_70k_reward_fn
from areal_vllm_amd.trainer.rl import run_grpo
from areal_vllm_amd.utils.hf_utils import load_hf_processor_and_tokenizer
from areal_vllm_amd.workflow.vision_rlvr import VisionRLVRWorkflow


def build(config, tokenizer, reward_fn, log_dir):
    processor, tokenizer = load_hf_processor_and_tokenizer(config.tokenizer_path)
    mk = lambda g, scope, sub: VisionRLVRWorkflow(  # noqa: E731
        reward_fn, g, tokenizer, processor, enable_thinking=False, rollout_stat_scope=scope,
        dump_dir=os.path.join(log_dir, sub))
    return (mk(config.gconfig, "rollout", "generated"),
            mk(config.gconfig.new(temperature=0.6), "eval-rollout", "generated-eval"))


def main(args):
    run_grpo(args, reward_fn=clevr_count_70k_reward_fn, workflow_builder=build)


if __name__ == "__main__":
    main(sys.argv[1:])
