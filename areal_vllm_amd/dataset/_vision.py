"""Shared pieces of the vision-language datasets: image normalisation and chat formatting."""
import math
from io import BytesIO
from typing import Any, Dict, List, Union


def to_rgb_image(image: Union[Dict[str, Any], "Image.Image", bytes], max_pixels: int = 512 * 512,
                 as_png_bytes: bool = False):
    """Decode, convert to RGB and shrink to at most `max_pixels`; optionally re-encode as PNG bytes."""
    from PIL import Image

    if isinstance(image, dict) and "bytes" in image:
        image = Image.open(BytesIO(image["bytes"]))
    elif isinstance(image, (bytes, bytearray)):
        image = Image.open(BytesIO(image))
    if image.mode != "RGB":
        image = image.convert("RGB")
    w, h = image.size
    if w * h > max_pixels:
        f = math.sqrt(max_pixels / (w * h))
        image = image.resize((max(1, int(w * f)), max(1, int(h * f))))
    if as_png_bytes:
        buf = BytesIO()
        image.save(buf, format="PNG")
        return buf.getvalue()
    return image


def image_token_of(processor) -> str:
    tok = getattr(processor, "image_token", None)
    return tok if tok is not None else "<|vision_start|><|image_pad|><|vision_end|>"


def chat_text(processor, question: str) -> str:
    messages = [{"role": "user", "content": question}]
    tokenizer = getattr(processor, "tokenizer", processor)
    return tokenizer.apply_chat_template(messages, add_generation_prompt=True, tokenize=False)


def shard(ds, rank: int, world_size: int):
    from datasets.distributed import split_dataset_by_node

    return split_dataset_by_node(ds, rank=rank, world_size=world_size)
