"""Tool-use environment interface for agentic workflows (``areal/api/env_api.py`` in the reference)."""
import abc
from typing import Any, Dict, List


class Environment(abc.ABC):
    async def ainitialize(self):
        """Acquire resources (sandboxes, connections)."""

    def list_tools(self) -> List[Dict[str, Any]]:
        """OpenAI-style tool schemas the agent may call."""
        return []

    @abc.abstractmethod
    async def aexecute(self, tool_name: str, tool_args: Dict[str, Any]) -> Any:
        """Run one tool call and return its observation."""

    async def aclose(self):
        """Release resources."""
