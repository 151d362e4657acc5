"""CLEVR-Count-70k vision-language dataset (counting objects in rendered scenes).

SFT rows: processor-tokenised prompt+answer with a loss mask on the answer and the pixel tensors; RL
rows: chat text with the image placeholder, the images, and the numeric answer. Field names follow
``areal/dataset/clevr_count_70k.py:40-163``.
"""
import re
from typing import Optional

from . import load_raw_dataset
from ._vision import chat_text, image_token_of, shard, to_rgb_image

PROMPT_TAIL = " Give your final count as a number in square brackets, e.g. [3]."


def _question(sample, processor) -> str:
    q = sample["problem"].replace("<image>", image_token_of(processor))
    return q + PROMPT_TAIL


def _answer(sample) -> str:
    m = re.findall(r"-?\d+", str(sample["answer"]))
    return m[-1] if m else str(sample["answer"]).strip()


def get_clevr_count_70k_sft_dataset(path: str, split: str, processor, rank: int, world_size: int,
                                    max_length: Optional[int] = None):
    ds = load_raw_dataset(path, split)
    tokenizer = processor.tokenizer

    def process(sample):
        images = [to_rgb_image(im) for im in sample["images"]]
        prompt = chat_text(processor, _question(sample, processor))
        answer = f"[{_answer(sample)}]" + tokenizer.eos_token
        enc = processor(text=[prompt + answer], images=images, padding=False, return_tensors="pt")
        ids = enc["input_ids"][0].tolist()
        n_prompt = len(processor(text=[prompt], images=images, padding=False, return_tensors="pt")["input_ids"][0])
        return {"input_ids": ids, "loss_mask": [0] * n_prompt + [1] * (len(ids) - n_prompt),
                "pixel_values": enc["pixel_values"], "image_grid_thw": enc["image_grid_thw"]}

    ds = ds.map(process).remove_columns(["images", "problem", "answer"])
    if max_length is not None:
        ds = ds.filter(lambda x: len(x["input_ids"]) <= max_length)
    return shard(ds, rank, world_size)


def get_clevr_count_70k_rl_dataset(path: str, split: str, processor, rank: int, world_size: int,
                                   max_length: Optional[int] = None):
    ds = load_raw_dataset(path, split)

    def process(sample):
        return {"messages": chat_text(processor, _question(sample, processor)),
                "images": [to_rgb_image(im, as_png_bytes=True) for im in sample["images"]],
                "answer": _answer(sample)}

    ds = ds.map(process).remove_columns(["problem"])
    if max_length is not None:
        def short_enough(sample):
            from ._vision import to_rgb_image as dec

            enc = processor(text=[sample["messages"]], images=[dec(b) for b in sample["images"]], padding=False,
                            return_tensors="pt")
            return enc["input_ids"].shape[1] <= max_length

        ds = ds.filter(short_enough)
    return shard(ds, rank, world_size)
