"""Tool-call extraction from generated text.

The reference delegates to SGLang's ``FunctionCallParser`` (``areal/experimental/openai/
tool_call_parser.py:17-55``); this is a self-contained parser for the formats its configs name:

* ``qwen25`` / ``qwen`` / ``hermes``: ``<tool_call>{"name": ..., "arguments": {...}}</tool_call>``
* ``llama3``: ``<|python_tag|>{"name": ..., "parameters": {...}}`` (several separated by ``;``)
* ``mistral``: ``[TOOL_CALLS] [{"name": ..., "arguments": {...}}, ...]``

Calls naming a function that is not in `tools` are dropped. Returns (tool_calls | None, remaining text,
finish_reason) with ``finish_reason`` switched from ``stop`` to ``tool_calls`` when a call was found.
"""
import json
import re
import uuid
from typing import Any, Dict, List, Optional, Tuple

from ...utils import logging
from .types import ChatCompletionMessageFunctionToolCall, Function

logger = logging.getLogger("tool-call-parser")

_QWEN = re.compile(r"<tool_call>\s*(.*?)\s*</tool_call>", re.S)


def _tool_names(tools: List[Any]) -> set:
    names = set()
    for t in tools or []:
        fn = t.get("function", t) if isinstance(t, dict) else getattr(t, "function", t)
        name = fn.get("name") if isinstance(fn, dict) else getattr(fn, "name", None)
        if name:
            names.add(name)
    return names


def _loads_many(blob: str) -> List[Dict[str, Any]]:
    """Parse one JSON object, a JSON array of objects, or several objects separated by ';' / newlines."""
    blob = blob.strip()
    try:
        v = json.loads(blob)
        return v if isinstance(v, list) else [v]
    except json.JSONDecodeError:
        pass
    out, dec, i = [], json.JSONDecoder(), 0
    while i < len(blob):
        j = blob.find("{", i)
        if j < 0:
            break
        try:
            obj, end = dec.raw_decode(blob, j)
            out.append(obj)
            i = end
        except json.JSONDecodeError:
            i = j + 1
    return out


def detect_format(text: str, tool_call_parser: Optional[str]) -> Optional[str]:
    p = (tool_call_parser or "").lower()
    if p in ("qwen25", "qwen", "qwen3", "hermes"):
        return "qwen" if "<tool_call>" in text else None
    if p == "llama3":
        return "llama3" if "<|python_tag|>" in text else None
    if p == "mistral":
        return "mistral" if "[TOOL_CALLS]" in text else None
    for tag, kind in (("<tool_call>", "qwen"), ("<|python_tag|>", "llama3"), ("[TOOL_CALLS]", "mistral")):
        if tag in text:
            return kind
    return None


def process_tool_calls(text: str, tools: List[Any], tool_call_parser: Optional[str], finish_reason: str
                       ) -> Tuple[Optional[List[ChatCompletionMessageFunctionToolCall]], str, str]:
    kind = detect_format(text, tool_call_parser)
    if kind is None:
        return None, text, finish_reason
    if finish_reason == "stop":
        finish_reason = "tool_calls"
    try:
        if kind == "qwen":
            blobs = _QWEN.findall(text)
            rest = _QWEN.sub("", text)
            # an unterminated trailing call (generation stopped at the closing tag)
            if "<tool_call>" in rest:
                head, _, tail = rest.partition("<tool_call>")
                blobs.append(tail)
                rest = head
            raw = [o for b in blobs for o in _loads_many(b)]
        elif kind == "llama3":
            rest, _, tail = text.partition("<|python_tag|>")
            raw = _loads_many(tail)
        else:
            rest, _, tail = text.partition("[TOOL_CALLS]")
            raw = _loads_many(tail)
        allowed = _tool_names(tools)
        calls = []
        for o in raw:
            if not isinstance(o, dict) or "name" not in o:
                continue
            if allowed and o["name"] not in allowed:
                logger.warning(f"model called undefined function {o['name']}")
                continue
            args = o.get("arguments", o.get("parameters", {}))
            calls.append(ChatCompletionMessageFunctionToolCall(
                id=f"call_{uuid.uuid4().hex[:24]}",
                function=Function(name=o["name"], arguments=args if isinstance(args, str) else json.dumps(args))))
        return (calls or None), rest.strip(), finish_reason
    except Exception as e:  # parsing must never fail the request
        logger.error(f"Tool call parsing error: {e}")
        return None, text, finish_reason
