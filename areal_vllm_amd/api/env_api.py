"""Tool-use environment interface for agentic workflows (``areal/api/env_api.py`` in the reference)."""
import abc
from typing import Any, Dict, List, Optional


class Environment(abc.ABC):
    """A set of tools an agent can call during an episode. Usable as ``async with env: ...``; only
    `aexecute` must be provided."""

    tools: Optional[List[Dict[str, Any]]] = None  # subclasses may simply set their OpenAI-style schemas here

    @abc.abstractmethod
    async def aexecute(self, tool_name: str, tool_args: Dict[str, Any]) -> Any:
        """Run one tool call and return its observation."""

    def list_tools(self) -> List[Dict[str, Any]]:
        """OpenAI-style tool schemas the agent may call."""
        return list(self.tools or ())

    def has_tool(self, tool_name: str) -> bool:
        return any(t.get("function", t).get("name") == tool_name for t in self.list_tools())

    async def ainitialize(self) -> None:
        """Acquire resources (sandboxes, connections)."""
        return None

    async def aclose(self) -> None:
        """Release resources."""
        return None

    async def __aenter__(self) -> "Environment":
        await self.ainitialize()
        return self

    async def __aexit__(self, exc_type, exc, tb) -> bool:
        await self.aclose()
        return False
