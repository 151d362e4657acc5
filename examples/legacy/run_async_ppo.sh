#!/bin/bash
# Asynchronous PPO with the worker runtime on one 8-GPU node: 4 generation servers + trainer dp 2 x pp 2.
python3 -m areal_vllm_amd.apps.quickstart async-ppo-math \
    allocation_mode=sglang.d4p1m1+d2p2m1 \
    fileroot=experiments \
    actor.path=Qwen/Qwen3-1.7B \
    ref.path=Qwen/Qwen3-1.7B ref.backend=inference \
    dataset.args.dataset_path=data/boba_106k_0319.jsonl \
    env.args.dataset_path=data/boba_106k_0319.jsonl \
    train_bs_n_seqs=32 \
    group_size=8 \
    ppo.gen.max_new_tokens=4096 \
    ppo.ppo_n_minibatches=4 \
    mb_spec.max_tokens_per_mb=32768 \
    max_concurrent_rollouts=16 \
    max_head_offpolicyness=4 \
    mode=local "$@"
