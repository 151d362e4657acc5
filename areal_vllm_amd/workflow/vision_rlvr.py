"""Single-turn RLVR for vision-language models: the RLVR episode with an image-aware request.

The HF processor renders (chat text, images) into token ids with one placeholder per merged patch. Remote
servers receive the images themselves (base64 in ``ModelRequest.image_data``), an in-process engine the
processed patches (request metadata); every returned row carries ``multi_modal_input`` for the trainer's
vision tower. Counterpart of ``areal/workflow/vision_rlvr.py``.
"""
import uuid
from typing import Any, Dict, Optional

from ..api.cli_args import GenerationHyperparameters
from ..api.io_struct import ModelRequest
from ..utils.image import image2base64
from .rlvr import RLVRWorkflow


class VisionRLVRWorkflow(RLVRWorkflow):
    def __init__(self, reward_fn, gconfig: GenerationHyperparameters, tokenizer, processor,
                 enable_thinking: bool = False, rollout_stat_scope: str = "rollout", dump_dir: Optional[str] = None,
                 reward_in_process: bool = False):
        super().__init__(reward_fn, gconfig, tokenizer, enable_thinking, rollout_stat_scope=rollout_stat_scope,
                         dump_dir=dump_dir, reward_in_process=reward_in_process)
        self.processor = processor

    def prepare(self, data: Dict[str, Any]):
        from ..dataset._vision import to_rgb_image

        pictures = [to_rgb_image(x) if isinstance(x, (bytes, bytearray, dict)) else x for x in data["images"]]
        enc = self.processor(images=pictures, text=data["messages"], padding=False, return_tensors="pt")
        patches = {"pixel_values": enc["pixel_values"], "image_grid_thw": enc["image_grid_thw"]}
        req = ModelRequest(rid=uuid.uuid4().hex, input_ids=enc["input_ids"][0].tolist(), image_data=image2base64(pictures),
                           gconfig=self.gconfig.new(n_samples=1), tokenizer=self.tokenizer, processor=self.processor,
                           metadata={"multi_modal_input": patches})
        return req, {"multi_modal_input": [patches]}
