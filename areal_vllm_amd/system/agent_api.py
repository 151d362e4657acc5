"""Agent / environment-service interfaces of the rollout workers and their registries.

An agent turns one prompt into zero or more trajectories by exchanging observations (generation
requests: ``(qid, token_ids, gconfig)`` on `obs_queue`) and actions (``BundledGenerationOutputs`` on
`act_queue`) with the worker's inference client, and by stepping an environment that scores the answers.
Contracts of ``realhf/api/core/agent_api.py:15-42`` and ``realhf/api/core/env_api.py``.
"""
import abc
import asyncio
import dataclasses
from typing import Any, Dict, List, Optional, Tuple

from .data_api import SequenceSample


@dataclasses.dataclass
class AgentAbstraction:
    type_: str = "null"
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class EnvServiceAbstraction:
    type_: str = "null"
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)


class Agent(abc.ABC):
    async def collect_trajectory(self, prompt: SequenceSample, env: "EnvironmentService", obs_queue: asyncio.Queue,
                                 act_queue: asyncio.Queue) -> List[SequenceSample]:
        raise NotImplementedError()


class EnvironmentService(abc.ABC):
    async def reset(self, seed: Optional[int] = None, options: Optional[Dict] = None) -> Tuple[Any, Dict]:
        return None, {}

    async def step(self, action: Any) -> Tuple[Any, Any, bool, bool, Dict]:
        """-> (observation, reward, terminated, truncated, info)."""
        raise NotImplementedError()


class NullEnvironment(EnvironmentService):
    async def step(self, action):
        return None, 0.0, True, False, {}


ALL_AGENTS: Dict[str, type] = {}
ALL_ENVS: Dict[str, type] = {"null": NullEnvironment}


def register_agent(name: str, cls):
    assert name not in ALL_AGENTS, name
    ALL_AGENTS[name] = cls


def register_environment(name: str, cls):
    assert name not in ALL_ENVS, name
    ALL_ENVS[name] = cls


def make_agent(cfg: AgentAbstraction) -> Agent:
    from ..impl import agent as _a  # noqa: F401  (registers the built-in agents)

    return ALL_AGENTS[cfg.type_](**cfg.args)


def make_env(cfg: EnvServiceAbstraction) -> EnvironmentService:
    from ..impl import environment as _e  # noqa: F401

    return ALL_ENVS[cfg.type_](**cfg.args)
