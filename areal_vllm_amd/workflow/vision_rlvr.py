"""Single-turn RLVR for vision-language models.

The processor turns (chat text, images) into token ids with image placeholders; images travel to the
rollout engine base64-encoded in ``ModelRequest.image_data``; every returned row carries
``multi_modal_input = [{pixel_values, image_grid_thw}]`` for the trainer. Behaviour of
``areal/workflow/vision_rlvr.py:22-144``.
"""
import asyncio
import os
import uuid
from typing import Any, Dict, Optional

import torch

from ..api.cli_args import GenerationHyperparameters
from ..api.io_struct import ModelRequest
from ..utils import stats_tracker
from ..utils.data import concat_padded_tensors
from ..utils.image import image2base64
from .rlvr import RLVRWorkflow


class VisionRLVRWorkflow(RLVRWorkflow):
    def __init__(self, reward_fn, gconfig: GenerationHyperparameters, tokenizer, processor,
                 enable_thinking: bool = False, rollout_stat_scope: str = "rollout", dump_dir: Optional[str] = None,
                 reward_in_process: bool = False):
        super().__init__(reward_fn, gconfig, tokenizer, enable_thinking, rollout_stat_scope=rollout_stat_scope,
                         dump_dir=dump_dir, reward_in_process=reward_in_process)
        self.processor = processor

    async def arun_episode(self, engine, data: Dict[str, Any]):
        from ..dataset._vision import to_rgb_image

        images = [to_rgb_image(im) if isinstance(im, (bytes, bytearray, dict)) else im for im in data["images"]]
        processed = self.processor(images=images, text=data["messages"], padding=False, return_tensors="pt")
        input_ids = processed["input_ids"].tolist()[0]
        n = self.gconfig.n_samples
        # remote servers get the images themselves (base64), an in-process engine the processed patches
        req = ModelRequest(rid=uuid.uuid4().hex, input_ids=input_ids, image_data=image2base64(images),
                           gconfig=self.gconfig.new(n_samples=1), tokenizer=self.tokenizer, processor=self.processor,
                           metadata={"multi_modal_input": {"pixel_values": processed["pixel_values"],
                                                           "image_grid_thw": processed["image_grid_thw"]}})
        resps = await asyncio.gather(*[engine.agenerate(req.copy()) for _ in range(n)])
        version = engine.get_version()
        prompt_str = self.tokenizer.decode(input_ids)
        results, dump = [], []
        extra = {k: v for k, v in data.items() if k not in ("messages", "images")}
        for resp in resps:
            seq = resp.input_tokens + resp.output_tokens
            completion_str = self.tokenizer.decode(resp.output_tokens)
            reward = await self._reward(prompt_str, completion_str, resp.input_tokens, resp.output_tokens, extra)
            stats_tracker.get(self.rollout_stat_scope).scalar(reward=reward)
            results.append(dict(
                input_ids=torch.tensor(seq).unsqueeze(0),
                loss_mask=torch.tensor([0] * resp.input_len + [1] * resp.output_len).unsqueeze(0),
                multi_modal_input=[{"pixel_values": processed["pixel_values"],
                                    "image_grid_thw": processed["image_grid_thw"]}],
                logprobs=torch.tensor([0.0] * resp.input_len + resp.output_logprobs).unsqueeze(0),
                versions=torch.tensor([-1] * resp.input_len + resp.output_versions).unsqueeze(0),
                attention_mask=torch.ones(len(seq), dtype=torch.bool).unsqueeze(0),
                rewards=torch.tensor([float(reward)]),
            ))
            dump.append((len(seq), reward, completion_str))
        if self.dump_dir is not None:
            d = os.path.join(self.dump_dir, str(version))
            os.makedirs(d, exist_ok=True)
            qid = next((str(data[k]) for k in ("query_id", "id", "qid") if k in data), uuid.uuid4().hex)
            with open(os.path.join(d, f"{qid}.txt"), "a") as f:
                for i, (sl, r, c) in enumerate(dump):
                    f.write(f"idx: {i + 1} / {n}, seqlen: {sl}, reward is {r}.\nprompt is \n{prompt_str}\n"
                            f"sequence is: \n{c}\n")
        return concat_padded_tensors(results)
