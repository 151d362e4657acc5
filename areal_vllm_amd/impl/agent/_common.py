"""Trajectory assembly shared by the math agents."""
import json
import os
import time
from typing import List, Optional

import torch

from ...system.data_api import SequenceSample
from ...system.model_api import BundledGenerationOutputs


def decode(tokenizer, ids: List[int]) -> str:
    if tokenizer is None:
        return " ".join(map(str, ids))
    return tokenizer.decode(ids, clean_up_tokenization_spaces=False, skip_special_tokens=True)


def group_sample(qid, prompt_ids: List[int], act: BundledGenerationOutputs, rewards: List[float],
                 birth_time: int, task_id: int = 0) -> SequenceSample:
    """One item holding the whole answer group of a prompt: packed sequences, prompt mask, behaviour
    log-probs ("short": one less than tokens), per-answer reward / eos flag / version range."""
    n = len(act.seqs)
    ids = torch.tensor([t for s in act.seqs for t in s], dtype=torch.long)
    pmask = torch.tensor([m for s in act.seqs for m in ([1] * act.prompt_len + [0] * (len(s) - act.prompt_len))],
                         dtype=torch.bool)
    logp = torch.tensor([x for lp in act.logprobs for x in lp], dtype=torch.float32)
    data = dict(
        packed_input_ids=ids, prompt_mask=pmask, packed_logprobs=logp,
        seq_no_eos_mask=torch.tensor(act.no_eos, dtype=torch.bool),
        packed_prompts=torch.tensor(prompt_ids, dtype=torch.long),
        version_start=torch.tensor(act.version_start, dtype=torch.int), version_end=torch.tensor(act.version_end, dtype=torch.int),
        rewards=torch.tensor(rewards, dtype=torch.float32), birth_time=torch.tensor([birth_time], dtype=torch.long),
        task_ids=torch.tensor([task_id], dtype=torch.long),
    )
    seqlens = dict(
        packed_input_ids=[[len(s) for s in act.seqs]], prompt_mask=[[len(s) for s in act.seqs]],
        packed_logprobs=[[len(s) - 1 for s in act.seqs]], seq_no_eos_mask=[[1] * n], packed_prompts=[[len(prompt_ids)]],
        version_start=[[1] * n], version_end=[[1] * n], rewards=[[1] * n], birth_time=[[1]], task_ids=[[1]],
    )
    return SequenceSample(keys=set(data), ids=[qid], seqlens=seqlens, data=data)


def log_rewards_to_file(save_path: Optional[str], qid: str, prompt: str, answers: List[str], seqlens: List[int],
                        prompt_len: int, rewards: List[float], success: List[bool], version_starts: List[int],
                        version_ends: List[int]) -> None:
    if not save_path:
        return
    group = str(version_starts[0])
    d = os.path.join(save_path, group)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"{qid}.txt"), "a") as f:
        for i, (a, r, ok, sl) in enumerate(zip(answers, rewards, success, seqlens)):
            f.write(f"idx: {i + 1} / {len(answers)}, seqlen: {sl}, head version: {version_starts[i]}, tail version: "
                    f"{version_ends[i]}.\nreward is {r}, prompt is {prompt}\nsequence is: {a}\n")
    with open(os.path.join(d, f"{qid}.jsonl"), "a") as f:
        for i in range(len(answers)):
            f.write(json.dumps(dict(prompt=prompt, prompt_len=prompt_len, answer=answers[i], seqlen=seqlens[i],
                                    reward=rewards[i], success=bool(success[i]), version_start=version_starts[i],
                                    version_end=version_ends[i], time=time.time())) + "\n")
