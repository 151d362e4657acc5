"""Conversions between the worker runtime's packed `SequenceSample` and the padded ``[B, S]`` batches the
engines consume.

Runtime layout (one item = one prompt with its group of answers): ``packed_input_ids`` holds the
sequences of the group back to back, "short" per-token arrays (``packed_logprobs``, ``packed_ref_logprobs``,
``proximal_logprobs``) have one entry less per sequence (entry i belongs to token i+1), per-answer scalars
(``rewards``, ``seq_no_eos_mask``, ``version_start`` ...) have one entry per sequence.
Padded layout: token-aligned rows; log-prob of token t sits at position t (position 0 is zero).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ...system.data_api import SequenceSample
from ...utils.datapack import flat2d


def seqlens_of(sample: SequenceSample, key: str = "packed_input_ids") -> List[int]:
    return [int(x) for x in flat2d(sample.seqlens[key])]


def group_sizes(sample: SequenceSample, key: str = "packed_input_ids") -> List[int]:
    return [len(l) for l in sample.seqlens[key]]


def pad_sequences(flat: torch.Tensor, lens: Sequence[int], width: int, shift: int = 0) -> torch.Tensor:
    """Rows of `flat` (concatenated pieces of `lens`) placed at column offset `shift` of a [B, width] tensor."""
    out = flat.new_zeros((len(lens), width) + tuple(flat.shape[1:]))
    off = 0
    for b, n in enumerate(lens):
        out[b, shift:shift + n] = flat[off:off + n]
        off += n
    return out


def to_padded_batch(sample: SequenceSample, device, short_keys: Sequence[str] = (),
                    seq_keys: Sequence[str] = (), token_keys: Sequence[str] = ()) -> Dict[str, torch.Tensor]:
    lens = seqlens_of(sample)
    S = max(lens)
    ids = sample.data["packed_input_ids"].to(device)
    batch = dict(input_ids=pad_sequences(ids, lens, S).long())
    am = torch.zeros(len(lens), S, dtype=torch.bool, device=device)
    for b, n in enumerate(lens):
        am[b, :n] = True
    batch["attention_mask"] = am
    for k in token_keys:
        if k in sample.keys and sample.data.get(k) is not None:
            batch[k] = pad_sequences(sample.data[k].to(device), lens, S)
    for k in short_keys:
        if k in sample.keys and sample.data.get(k) is not None:
            batch[k] = pad_sequences(sample.data[k].to(device).float(), [n - 1 for n in lens], S, shift=1)
    for k in seq_keys:
        if k in sample.keys and sample.data.get(k) is not None:
            batch[k] = sample.data[k].to(device)
    return batch


def short_from_padded(x: torch.Tensor, lens: Sequence[int]) -> torch.Tensor:
    """[B, S] with entry (b, t) = value for predicting token t+1  ->  packed short array."""
    return torch.cat([x[b, : n - 1] for b, n in enumerate(lens)]) if lens else x.new_zeros(0)


def result_sample(sample: SequenceSample, data: Dict[str, torch.Tensor], kinds: Dict[str, str]) -> SequenceSample:
    """Output sample with the ids / grouping of `sample`; `kinds[key]` in {"token", "short", "seq"}."""
    base = sample.seqlens["packed_input_ids"]
    sl = {}
    for k, kind in kinds.items():
        if kind == "token":
            sl[k] = [list(g) for g in base]
        elif kind == "short":
            sl[k] = [[n - 1 for n in g] for g in base]
        else:
            sl[k] = [[1] * len(g) for g in base]
    return SequenceSample(keys=set(data), ids=list(sample.ids), seqlens=sl, data={k: v.cpu() for k, v in data.items()})
