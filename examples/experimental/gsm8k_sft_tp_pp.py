"""SFT on GSM8K with a TP x PP trainer (the reference's gsm8k_sft_megatron.py).

python -m areal_vllm_amd.launcher.local examples/experimental/gsm8k_sft_tp_pp.py --config examples/experimental/configs/gsm8k_sft_tp_pp.yaml
"""
import sys

from areal_vllm_amd.trainer.sft import run_sft


def main(args):
    run_sft(args)


if __name__ == "__main__":
    main(sys.argv[1:])
