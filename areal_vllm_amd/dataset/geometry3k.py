"""Geometry3K vision-language dataset (diagram + question -> short answer).

Same row schema as the CLEVR dataset; the answer is free-form (number, expression) and must be given in
square brackets. Reference: ``areal/dataset/geometry3k.py:40-179``.
"""
from typing import Optional

from . import load_raw_dataset
from ._vision import chat_text, image_token_of, shard, to_rgb_image

PROMPT_TAIL = " Think step by step, then give the final answer in square brackets, e.g. [12.5]."


def _question(sample, processor) -> str:
    return sample["problem"].replace("<image>", image_token_of(processor)) + PROMPT_TAIL


def get_geometry3k_sft_dataset(path: str, split: str, processor, rank: int, world_size: int,
                               max_length: Optional[int] = None):
    ds = load_raw_dataset(path, split)
    tokenizer = processor.tokenizer

    def process(sample):
        images = [to_rgb_image(im, 448 * 448) for im in sample["images"]]
        prompt = chat_text(processor, _question(sample, processor))
        answer = f"[{str(sample['answer']).strip()}]" + tokenizer.eos_token
        enc = processor(text=[prompt + answer], images=images, padding=False, return_tensors="pt")
        ids = enc["input_ids"][0].tolist()
        n_prompt = len(processor(text=[prompt], images=images, padding=False, return_tensors="pt")["input_ids"][0])
        return {"input_ids": ids, "loss_mask": [0] * n_prompt + [1] * (len(ids) - n_prompt),
                "pixel_values": enc["pixel_values"], "image_grid_thw": enc["image_grid_thw"]}

    ds = ds.map(process).remove_columns(["images", "problem", "answer"])
    if max_length is not None:
        ds = ds.filter(lambda x: len(x["input_ids"]) <= max_length)
    return shard(ds, rank, world_size)


def get_geometry3k_rl_dataset(path: str, split: str, processor, rank: int, world_size: int,
                              max_length: Optional[int] = None):
    ds = load_raw_dataset(path, split)

    def process(sample):
        return {"messages": chat_text(processor, _question(sample, processor)),
                "images": [to_rgb_image(im, 448 * 448, as_png_bytes=True) for im in sample["images"]],
                "answer": str(sample["answer"]).strip()}

    ds = ds.map(process).remove_columns(["problem"])
    if max_length is not None:
        ds = ds.filter(lambda s: len(processor.tokenizer.encode(s["messages"])) <= max_length)
    return shard(ds, rank, world_size)
