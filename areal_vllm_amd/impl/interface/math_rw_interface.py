"""Rule-based multi-task reward "model": decodes every answer, routes it to the verifier of its task
(math answer checking, code unit tests; locally or through the function-call service) and returns one
reward per answer. Needs no parameters, so it is hosted with the `null` backend. Behaviour of
``realhf/impl/model/interface/math_rw_interface.py:181-518`` (MultiTaskRewardInterface)."""
import asyncio
import json
import os
from typing import Dict, List, Optional

import torch

from ...impl.environment.math_code_single_step_env import code_verify_call, extract_code, load_metadata, \
    math_verify_call
from ...system.data_api import SequenceSample
from ...system.model_api import Model, ModelInterface, register_interface
from ...utils import logging
from ._convert import seqlens_of

logger = logging.getLogger("reward-interface")


class MultiTaskRewardInterface(ModelInterface):
    def __init__(self, dataset_path: str = "", tokenizer_path: Optional[str] = None, id2info: Optional[Dict] = None,
                 output_scaling: float = 1.0, output_bias: float = 0.0, rw_type: str = "sparse",
                 check_xml_format: bool = False, group_size: int = 1, answer_save_path: Optional[str] = None,
                 check_verifier_status: bool = False, **_):
        self.id2info = id2info if id2info is not None else (load_metadata(dataset_path)[0] if dataset_path else {})
        self.tokenizer_path = tokenizer_path
        self.output_scaling, self.output_bias = output_scaling, output_bias
        self.rw_type = rw_type
        self.check_xml_format = check_xml_format
        self.answer_save_path = answer_save_path
        self._tok = None

    def _decode(self, model: Model, ids: List[int]) -> str:
        tok = model.tokenizer or self._tok
        if tok is None and self.tokenizer_path:
            from ...utils.hf_utils import load_hf_tokenizer

            tok = self._tok = load_hf_tokenizer(self.tokenizer_path)
        if tok is None:
            return " ".join(map(str, ids))
        return tok.decode(ids, clean_up_tokenization_spaces=False, skip_special_tokens=True)

    @staticmethod
    def _xml_ok(text: str) -> bool:
        return text.count("<think>") == 1 and text.count("</think>") == 1 and \
            text.index("<think>") < text.index("</think>")

    def inference(self, model: Model, data: SequenceSample, mb_spec) -> Optional[SequenceSample]:
        ids = data.data["packed_input_ids"].tolist()
        pm = data.data["prompt_mask"].tolist() if "prompt_mask" in data.keys else None
        answers, qids, tasks = [], [], []
        off = 0
        for item_id, group in zip(data.ids, data.seqlens["packed_input_ids"]):
            qid = str(item_id).split("@")[0]
            for n in group:
                seq = ids[off:off + n]
                if pm is not None:
                    seq = [t for t, m in zip(seq, pm[off:off + n]) if not m]
                answers.append(self._decode(model, seq))
                qids.append(qid)
                tasks.append(self.id2info.get(qid, {}).get("task", "math"))
                off += n
        scores = [0.0] * len(answers)
        for task in sorted(set(tasks)):
            idx = [i for i, t in enumerate(tasks) if t == task]
            if task == "math":
                r = math_verify_call(self.id2info, [answers[i] for i in idx], [qids[i] for i in idx])
            elif task == "code":
                r = code_verify_call(self.id2info, [extract_code(answers[i]) or "" for i in idx], [qids[i] for i in idx])
            else:
                raise NotImplementedError(f"no verifier for task '{task}'")
            for i, x in zip(idx, r):
                scores[i] = float(x)
        if self.check_xml_format:
            scores = [s if self._xml_ok(a) else 0.0 for s, a in zip(scores, answers)]
        rewards = torch.tensor([((s - 0.5) * 2 - self.output_bias) * self.output_scaling for s in scores])
        if self.answer_save_path:
            os.makedirs(self.answer_save_path, exist_ok=True)
            with open(os.path.join(self.answer_save_path, f"v{model.version.global_step}.jsonl"), "a") as f:
                for q, a, s in zip(qids, answers, scores):
                    f.write(json.dumps(dict(qid=q, answer=a, score=s)) + "\n")
        model.inc_version()
        sl = [[1] * len(g) for g in data.seqlens["packed_input_ids"]]
        return SequenceSample(keys={"rewards"}, ids=list(data.ids), seqlens=dict(rewards=sl), data=dict(rewards=rewards))


register_interface("rw-math-code", MultiTaskRewardInterface)
register_interface("rw_math_code", MultiTaskRewardInterface)
