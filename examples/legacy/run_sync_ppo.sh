#!/bin/bash
# Synchronous PPO: generation, reward, (reference) inference and training are steps of one data-flow graph on
# the same 8 model workers.
python3 -m areal_vllm_amd.apps.quickstart ppo-math \
    allocation_mode=d8p1m1 \
    fileroot=experiments \
    actor.path=Qwen/Qwen3-1.7B \
    dataset.args.dataset_path=data/boba_106k_0319.jsonl \
    train_bs_n_seqs=32 \
    group_size=8 \
    ppo.gen.max_new_tokens=4096 \
    ppo.ppo_n_minibatches=4 \
    mb_spec.max_tokens_per_mb=32768 \
    mode=local "$@"
