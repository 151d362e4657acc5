"""Math / code prompt dataset with verification metadata. Rows: {"query_id", "prompt" | "input_ids",
"task": "math"|"code", "solutions": [...] (math) or "input_output": json (code)}; rows whose metadata
is malformed are dropped at load time. ``id2info`` is what reward interfaces and environments look
answers up in. Behaviour of ``realhf/impl/dataset/math_code_dataset.py`` (check_math/code_metadata_entries,
MATHCodePromptDataset)."""
import json
from typing import Callable, Dict, List, Optional

from ...utils import logging
from . import register_dataset
from .prompt_dataset import PromptDataset

logger = logging.getLogger("math-code-dataset")


def check_math_metadata_entries(d: Dict) -> Dict:
    assert d.get("task") == "math" or "task" not in d
    assert "query_id" in d and ("prompt" in d or "input_ids" in d)
    assert isinstance(d["solutions"], list) and all(isinstance(s, str) for s in d["solutions"])
    d["query_id"] = str(d["query_id"])
    return d


def check_code_metadata_entries(d: Dict) -> Dict:
    assert d["task"] == "code"
    assert "query_id" in d and ("prompt" in d or "input_ids" in d)
    io = json.loads(d["input_output"]) if isinstance(d["input_output"], str) else d["input_output"]
    assert len(io["inputs"]) == len(io["outputs"])
    d["query_id"] = str(d["query_id"])
    return d


def load_metadata(path: str):
    id2info, cnt = {}, {}
    with open(path) as f:
        for line in f:
            if not line.strip():
                continue
            d = json.loads(line)
            d.setdefault("task", "math")
            try:
                d = check_code_metadata_entries(d) if d["task"] == "code" else check_math_metadata_entries(d)
            except Exception as e:
                logger.warning(f"dropping malformed row {d.get('query_id')}: {e!r}")
                continue
            id2info[d["query_id"]] = d
            cnt[d["task"]] = cnt.get(d["task"], 0) + 1
    return id2info, cnt


class MATHCodePromptDataset(PromptDataset):
    def __init__(self, seed: int, dp_rank: int, world_size: int, tokenizer=None, max_length: Optional[int] = None,
                 dataset_path: Optional[str] = None, dataset_builder: Optional[Callable[[], List[Dict]]] = None,
                 filter_threshold: float = 1e4, max_filter_percentage: float = 0.0):
        if dataset_path is not None:
            self.id2info, cnt = load_metadata(dataset_path)
            rows = list(self.id2info.values())
            builder = lambda: [dict(r) for r in rows]  # noqa: E731
            logger.info(f"math/code dataset: {cnt}")
        else:
            rows = dataset_builder()
            self.id2info = {str(r["query_id"]): r for r in rows}
            builder = lambda: [dict(r) for r in rows]  # noqa: E731
        super().__init__(seed, dp_rank, world_size, tokenizer, max_length, None, builder, filter_threshold,
                         max_filter_percentage)


register_dataset("math_code_prompt", MATHCodePromptDataset)
