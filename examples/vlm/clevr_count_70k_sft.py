"""Supervised fine-tuning of a vision-language model on CLEVR-Count-70k."""
import sys

from areal_vllm_amd.trainer.sft import run_sft


def main(args):
    run_sft(args, processor_needed=True)


if __name__ == "__main__":
    main(sys.argv[1:])
