"""Cluster description: node names, GPU model, shared file root. Loaded from a JSON file
(``{"cluster_type", "cluster_name", "fileroot", "gpu_type", "node_name_prefix", "n_nodes",
"n_gpus_per_node"}``) or taken from the experiment config - the reference does the same in
``realhf/base/cluster.py``; the default describes one 8 x MI355X node."""
import dataclasses
import json
import os
from typing import List, Optional


@dataclasses.dataclass
class ClusterSpec:
    cluster_type: str = "local"  # local | slurm | ray
    cluster_name: str = "local"
    fileroot: str = "/tmp/areal"
    gpu_type: str = "MI355X"
    gpu_memory_gb: int = 288
    node_name_prefix: str = "node"
    n_nodes: int = 1
    n_gpus_per_node: int = 8

    def node_names(self) -> List[str]:
        width = max(2, len(str(self.n_nodes)))
        return [f"{self.node_name_prefix}{i:0{width}d}" for i in range(1, self.n_nodes + 1)]

    @property
    def n_gpus(self) -> int:
        return self.n_nodes * self.n_gpus_per_node


def load_spec_from_file(path: str) -> ClusterSpec:
    with open(path) as f:
        d = json.load(f)
    known = {f.name for f in dataclasses.fields(ClusterSpec)}
    return ClusterSpec(**{k: v for k, v in d.items() if k in known})


def load_spec(config=None) -> ClusterSpec:
    """Precedence: `config.cluster.config_path` / env CLUSTER_SPEC_PATH (JSON file), else the fields of
    `config.cluster`, else the single-node default."""
    path = getattr(getattr(config, "cluster", None), "config_path", None) or os.environ.get("CLUSTER_SPEC_PATH")
    if path:
        return load_spec_from_file(path)
    c = getattr(config, "cluster", None)
    if c is None:
        return ClusterSpec()
    return ClusterSpec(cluster_name=getattr(c, "cluster_name", "local"), fileroot=getattr(c, "fileroot", "/tmp/areal"),
                       n_nodes=getattr(c, "n_nodes", 1), n_gpus_per_node=getattr(c, "n_gpus_per_node", 8))
