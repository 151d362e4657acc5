"""Data-flow graph of model function calls (MFCs).

An algorithm is a set of MFCs - (model, interface call type, input keys, output keys, batch size) -
whose dependencies follow from the keys: an edge A -> B exists when B consumes a key that A produces;
keys nobody produces come from the dataset. PPO is e.g. actor.generate -> {reward.inference,
ref.inference, critic.inference} -> {actor.train_step, critic.train_step}. Contract of
``realhf/api/core/dfg.py:56-330`` (MFCDef, build_graph, hooks).
"""
import dataclasses
import enum
from typing import Any, Dict, List, Optional, Set, Tuple

import networkx as nx

from ..api.cli_args import MicroBatchSpec


class ModelInterfaceType(str, enum.Enum):
    GENERATE = "generate"
    TRAIN_STEP = "train_step"
    EVALUATE = "evaluate"
    INFERENCE = "inference"


@dataclasses.dataclass(frozen=True)
class ModelName:
    role: str
    replica_id: int = 0

    def __str__(self):
        return f"{self.role}@{self.replica_id}"


@dataclasses.dataclass
class ModelInterfaceAbstraction:
    type_: str
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class OffloadHook:
    pass


@dataclasses.dataclass
class ParamReallocHook:
    """Move / broadcast parameters between two model replicas with different layouts (or EMA-mix them:
    target = eta * source + (1 - eta) * target)."""
    source: Optional[ModelName] = None
    target: Optional[ModelName] = None
    eta: float = 1.0


@dataclasses.dataclass
class MFCDef:
    name: str
    n_seqs: int
    interface_type: ModelInterfaceType
    interface_impl: ModelInterfaceAbstraction
    model_name: Any
    input_keys: Tuple[str, ...] = ()
    input_key_remap: Dict[str, str] = dataclasses.field(default_factory=dict)
    output_keys: Tuple[str, ...] = ()
    output_key_remap: Dict[str, str] = dataclasses.field(default_factory=dict)
    mb_spec: MicroBatchSpec = dataclasses.field(default_factory=MicroBatchSpec)
    min_n_seqs_per_pass: int = 1
    log_return_value: bool = False
    model_type: Optional[Any] = None
    model_path: Optional[str] = None
    _pre_hooks: List[Any] = dataclasses.field(default_factory=list)
    _post_hooks: List[Any] = dataclasses.field(default_factory=list)
    # filled by build_graph
    _G: Optional[nx.DiGraph] = None

    def __post_init__(self):
        if isinstance(self.model_name, str):
            self.model_name = ModelName(self.model_name, 0)
        self.input_keys, self.output_keys = tuple(self.input_keys), tuple(self.output_keys)

    @property
    def role(self) -> str:
        return self.model_name.role

    def add_pre_hook(self, h):
        if isinstance(h, OffloadHook):
            raise ValueError("Offload can only be a post hook")
        self._pre_hooks.append(h)

    def add_post_hook(self, h):
        self._post_hooks.append(h)

    @property
    def is_src(self) -> bool:
        return self._G is not None and self._G.in_degree(self.name) == 0

    @property
    def is_dst(self) -> bool:
        return self._G is not None and self._G.out_degree(self.name) == 0

    @property
    def parents(self) -> List["MFCDef"]:
        return [self._G.nodes[x]["object"] for x in self._G.predecessors(self.name)]

    @property
    def children(self) -> List["MFCDef"]:
        return [self._G.nodes[x]["object"] for x in self._G.successors(self.name)]

    @property
    def data_producers(self) -> Dict[str, Any]:
        return self._G.graph["data_producers"]

    @property
    def data_consumers(self) -> Dict[str, List[Any]]:
        return self._G.graph["data_consumers"]

    def is_dst_of_model_role(self, role: str) -> bool:
        """True when no descendant uses a model of the same role (the last call on that model in a step)."""
        return not any(self._G.nodes[d]["object"].role == role for d in nx.descendants(self._G, self.name))


def build_graph(nodes: List[MFCDef], verbose: bool = False) -> nx.DiGraph:
    G = nx.DiGraph()
    producers: Dict[str, Any] = {}
    consumers: Dict[str, List[Any]] = {}
    for n in nodes:
        assert n.name not in G, f"duplicate MFC name {n.name}"
        G.add_node(n.name, object=n)
        for k in n.output_keys:
            assert k not in producers, f"key {k} is produced by both {producers[k]} and {n.name}"
            producers[k] = n.name
    for n in nodes:
        for k in n.input_keys:
            consumers.setdefault(k, []).append(n.name)
            if k in producers and producers[k] != n.name:
                if G.has_edge(producers[k], n.name):
                    G[producers[k]][n.name]["keys"].append(k)
                else:
                    G.add_edge(producers[k], n.name, keys=[k])
    assert nx.is_directed_acyclic_graph(G), "the MFC graph has a cycle"
    G.graph["data_producers"] = {k: G.nodes[v]["object"].model_name for k, v in producers.items()}
    G.graph["data_consumers"] = {k: [G.nodes[v]["object"].model_name for v in vs] for k, vs in consumers.items()}
    G.graph["dataset_keys"] = sorted(k for k in consumers if k not in producers)
    for n in nodes:
        n._G = G
    return G


def topological_levels(G: nx.DiGraph) -> List[List[str]]:
    """MFC names grouped by depth: all calls of one level can run concurrently."""
    return [sorted(gen) for gen in nx.topological_generations(G)]
