#!/bin/bash
# Supervised fine-tuning on prompt / answer pairs with the worker runtime.
python3 -m areal_vllm_amd.apps.quickstart sft \
    allocation_mode=d8p1m1 \
    fileroot=experiments \
    model.path=Qwen/Qwen3-1.7B \
    dataset.args.dataset_path=data/sft_pairs.jsonl \
    train_bs_n_seqs=64 \
    mb_spec.max_tokens_per_mb=32768 \
    exp_ctrl.total_train_epochs=2 \
    mode=local "$@"
