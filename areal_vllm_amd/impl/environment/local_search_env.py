"""Offline retrieval environment for search-agent training: a tiny BM25 index over a local corpus with
the two tools such agents use (`search` for ranked snippets, `access` for a whole document). Stands in
for the web search / page access services of the reference's search-agent setup
(``examples/search-agent/local_1.5b_example.yaml``; the tool services themselves are external there)."""
import json
import math
import re
from collections import Counter
from typing import Any, Dict, List, Optional

from ...api.env_api import Environment

_TOK = re.compile(r"[a-z0-9]+")


class BM25Index:
    def __init__(self, docs: List[Dict[str, str]], k1: float = 1.5, b: float = 0.75):
        self.docs = docs
        self.k1, self.b = k1, b
        self.tfs = [Counter(_TOK.findall((d.get("title", "") + " " + d["text"]).lower())) for d in docs]
        self.lens = [sum(tf.values()) for tf in self.tfs]
        self.avg = max(1.0, sum(self.lens) / max(1, len(docs)))
        df = Counter()
        for tf in self.tfs:
            df.update(tf.keys())
        n = len(docs)
        self.idf = {t: math.log(1 + (n - c + 0.5) / (c + 0.5)) for t, c in df.items()}

    def search(self, query: str, topk: int = 3) -> List[int]:
        q = _TOK.findall(query.lower())
        scores = []
        for i, tf in enumerate(self.tfs):
            s = 0.0
            for t in q:
                f = tf.get(t, 0)
                if f:
                    s += self.idf.get(t, 0.0) * f * (self.k1 + 1) / (
                        f + self.k1 * (1 - self.b + self.b * self.lens[i] / self.avg))
            scores.append((s, i))
        scores.sort(key=lambda x: (-x[0], x[1]))
        return [i for s, i in scores[:topk] if s > 0]


class LocalSearchEnv(Environment):
    def __init__(self, index: BM25Index, topk: int = 3, snippet_chars: int = 240):
        self.index = index
        self.topk = topk
        self.snippet_chars = snippet_chars
        self.stats = dict(num_search_queries=0, num_success_search_queries=0, num_failed_search_queries=0,
                          num_pages=0, num_success_url_accesses=0, num_failed_url_accesses=0)

    @classmethod
    def from_jsonl(cls, path: str, **kw) -> "LocalSearchEnv":
        with open(path) as f:
            docs = [json.loads(l) for l in f if l.strip()]
        return cls(BM25Index(docs), **kw)

    def list_tools(self) -> List[Dict[str, Any]]:
        return [
            {"type": "function", "function": {
                "name": "search", "description": "Search the corpus; returns ranked snippets with their urls.",
                "parameters": {"type": "object", "properties": {"query": {"type": "string"}}, "required": ["query"]}}},
            {"type": "function", "function": {
                "name": "access", "description": "Return the full text of the document at `url`.",
                "parameters": {"type": "object", "properties": {"url": {"type": "string"}}, "required": ["url"]}}},
        ]

    def _url(self, i: int) -> str:
        return self.index.docs[i].get("url", f"doc://{i}")

    async def aexecute(self, tool_name: str, tool_args: Dict[str, Any]) -> Any:
        if tool_name == "search":
            self.stats["num_search_queries"] += 1
            hits = self.index.search(str(tool_args.get("query", "")), self.topk)
            if not hits:
                self.stats["num_failed_search_queries"] += 1
                return "no results"
            self.stats["num_success_search_queries"] += 1
            return "\n".join(f"[{r + 1}] {self._url(i)} {self.index.docs[i].get('title', '')}: "
                             f"{self.index.docs[i]['text'][: self.snippet_chars]}" for r, i in enumerate(hits))
        if tool_name == "access":
            url = str(tool_args.get("url", ""))
            for i in range(len(self.index.docs)):
                if self._url(i) == url:
                    self.stats["num_pages"] += 1
                    self.stats["num_success_url_accesses"] += 1
                    return self.index.docs[i]["text"]
            self.stats["num_failed_url_accesses"] += 1
            return f"cannot access {url}"
        raise ValueError(f"unknown tool {tool_name}")


def f1_answer_reward(prompt, completion, prompt_ids, completion_ids, answer: Optional[Any] = None, **kwargs) -> float:
    """Token-level F1 between the text inside the last <answer> block and the closest gold answer."""
    m = re.findall(r"<answer>(.*?)</answer>", completion or "", re.DOTALL)
    if not m or answer is None:
        return 0.0
    pred = _TOK.findall(m[-1].lower())
    golds = answer if isinstance(answer, (list, tuple)) else [answer]
    best = 0.0
    for g in golds:
        gold = _TOK.findall(str(g).lower())
        common = sum((Counter(pred) & Counter(gold)).values())
        if common == 0 or not pred or not gold:
            continue
        p, r = common / len(pred), common / len(gold)
        best = max(best, 2 * p * r / (p + r))
    return best
