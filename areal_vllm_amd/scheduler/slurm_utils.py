"""Slurm resource bookkeeping: what the cluster has free, and which task of which worker array goes to which node.

The worker runtime starts one srun step per worker type. Left to itself Slurm spreads the tasks of those steps
as it sees fit; for a GPU job that is wrong twice: model worker i must land next to model worker i+1 (tensor- and
pipeline-parallel neighbours talk over xGMI inside a node, only data-parallel traffic should cross nodes), and two
worker types that both want GPUs must not be promised the same devices. So the client asks the cluster what is
free (``scontrol show node``), packs every array onto nodes itself and pins the result with a hostfile
(``--distribution=arbitrary``) and a multi-prog file. Role of ``realhf/scheduler/slurm/utils.py:57-800``
(SlurmResource, available_hostnames, get_all_node_resources, allocate_resources, SlurmLaunchInfo.commit).
"""
import dataclasses
import re
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple


class SlurmResourceNotEnoughException(Exception):
    pass


@dataclasses.dataclass(frozen=True)
class NodeResource:
    cpu: int = 0
    mem: int = 0  # MB
    gpu: int = 0

    def __add__(self, o: "NodeResource") -> "NodeResource":
        return NodeResource(self.cpu + o.cpu, self.mem + o.mem, self.gpu + o.gpu)

    def __sub__(self, o: "NodeResource") -> "NodeResource":
        return NodeResource(self.cpu - o.cpu, self.mem - o.mem, self.gpu - o.gpu)

    def __mul__(self, k: int) -> "NodeResource":
        return NodeResource(self.cpu * k, self.mem * k, self.gpu * k)

    __rmul__ = __mul__

    def fits(self, o: "NodeResource") -> bool:
        """Can a request `o` be served from this much?"""
        return self.cpu >= o.cpu and self.mem >= o.mem and self.gpu >= o.gpu

    def valid(self) -> bool:
        return self.cpu >= 0 and self.mem >= 0 and self.gpu >= 0


_UNUSABLE = ("DOWN", "DRAIN", "DRAINED", "DRAINING", "FAIL", "FAILING", "MAINT", "NOT_RESPONDING", "INVAL", "RESERVED")


def _mem_mb(text: str) -> int:
    m = re.fullmatch(r"(\d+(?:\.\d+)?)([KMGT]?)", text.strip())
    if not m:
        return 0
    return int(float(m.group(1)) * {"": 1, "K": 1 / 1024, "M": 1, "G": 1024, "T": 1024 * 1024}[m.group(2)])


def _parse_tres(tres: str) -> NodeResource:
    """``cpu=128,mem=2000G,billing=128,gres/gpu=8`` (any order, typed gpus ``gres/gpu:mi355x=8`` included)."""
    cpu = mem = gpu = 0
    for item in filter(None, tres.split(",")):
        k, _, v = item.partition("=")
        if k == "cpu":
            cpu = int(v)
        elif k == "mem":
            mem = _mem_mb(v)
        elif k == "gres/gpu" or (k.startswith("gres/gpu:") and gpu == 0):
            gpu = int(v)
    return NodeResource(cpu, mem, gpu)


def parse_scontrol_nodes(text: str) -> Dict[str, NodeResource]:
    """Free resources per usable node from the output of ``scontrol show node``: configured minus allocated
    trackable resources; nodes that are down, draining, reserved or in maintenance are left out."""
    out: Dict[str, NodeResource] = {}
    for block in re.split(r"\n\s*\n", text.strip()):
        name = re.search(r"NodeName=(\S+)", block)
        state = re.search(r"State=(\S+)", block)
        cfg = re.search(r"CfgTRES=(\S*)", block)
        if not (name and state and cfg):
            continue
        flags = re.split(r"[+*~#$@^-]", state.group(1).upper())
        if any(f in _UNUSABLE for f in flags):
            continue
        alloc = re.search(r"AllocTRES=(\S*)", block)
        free = _parse_tres(cfg.group(1)) - _parse_tres(alloc.group(1) if alloc else "")
        if free.valid():
            out[name.group(1)] = free
    return out


def get_all_node_resources() -> Dict[str, NodeResource]:
    return parse_scontrol_nodes(subprocess.check_output(["scontrol", "show", "node"]).decode())


def expand_nodelist(spec: Optional[str]) -> Optional[List[str]]:
    """``a[01-03,07],b2`` -> [a01, a02, a03, a07, b2] (the compressed host lists of Slurm); None stays None."""
    if not spec:
        return None
    hosts: List[str] = []
    for part in re.findall(r"[^,\[]+(?:\[[^\]]*\])?", spec):
        m = re.fullmatch(r"([^\[]+)\[([^\]]*)\]", part)
        if not m:
            hosts.append(part)
            continue
        for rng in m.group(2).split(","):
            lo, _, hi = rng.partition("-")
            for i in range(int(lo), int(hi or lo) + 1):
                hosts.append(f"{m.group(1)}{str(i).zfill(len(lo))}")
    return hosts


@dataclasses.dataclass
class LaunchRequest:
    """One worker array: `n_tasks` identical tasks, each needing `per_task`."""
    name: str
    n_tasks: int
    per_task: NodeResource
    nodelist: Optional[str] = None
    exclude: Optional[str] = None


@dataclasses.dataclass
class Placement:
    name: str
    hosts: List[str]  # host of task i
    local_ids: List[int]  # index of task i among the tasks of this array on its host (its device ordinal)

    def hostfile(self) -> str:
        return "\n".join(self.hosts) + "\n"

    def nodelist(self) -> str:
        return ",".join(sorted(set(self.hosts), key=self.hosts.index))

    def multiprog(self, cmd: str) -> str:
        """Multi-prog configuration: task i runs `cmd` with {index} = i and {local} = its ordinal on the node."""
        return "\n".join(f"{i} {cmd.replace('{index}', str(i)).replace('{local}', str(l))}"
                         for i, l in enumerate(self.local_ids)) + "\n"


def allocate_resources(requests: Sequence[LaunchRequest], nodes: Dict[str, NodeResource]
                       ) -> Tuple[List[Placement], Dict[str, NodeResource]]:
    """Place every task of every request; returns the placements (in the order of `requests`) and what is left
    of the nodes. GPU arrays are placed first, largest first, and each is PACKED: consecutive tasks fill a node
    before the next node is opened, nodes with the most free GPUs first, so that consecutive worker indices
    (the tensor / pipeline neighbours of the rank grid) share a node. CPU-only arrays then go wherever room is
    left, least-loaded node first (masters, rollout workers and controllers stay off each other's cores).
    Raises `SlurmResourceNotEnoughException` naming the array that does not fit."""
    free = dict(nodes)
    placed: Dict[str, Placement] = {}
    order = sorted(requests, key=lambda r: (-(r.per_task.gpu > 0), -r.per_task.gpu * r.n_tasks, -r.n_tasks))
    for req in order:
        allowed = expand_nodelist(req.nodelist)
        banned = set(expand_nodelist(req.exclude) or [])
        cand = [h for h in free if (allowed is None or h in allowed) and h not in banned]
        hosts: List[str] = []
        local: List[int] = []
        remaining = req.n_tasks
        if req.per_task.gpu > 0:
            cand.sort(key=lambda h: (-free[h].gpu, h))
        else:
            cand.sort(key=lambda h: (-(free[h].cpu - 0), h))
        for h in cand:
            k = 0
            while remaining > 0 and free[h].fits(req.per_task):
                free[h] = free[h] - req.per_task
                hosts.append(h)
                local.append(k)
                k += 1
                remaining -= 1
                if req.per_task.gpu == 0 and k >= max(1, (req.n_tasks + len(cand) - 1) // len(cand)):
                    break  # CPU-only arrays spread evenly
            if remaining == 0:
                break
        if remaining > 0 and req.per_task.gpu == 0:  # a second pass without the even-spread cap
            for h in cand:
                k = hosts.count(h)
                while remaining > 0 and free[h].fits(req.per_task):
                    free[h] = free[h] - req.per_task
                    hosts.append(h)
                    local.append(k)
                    k += 1
                    remaining -= 1
        if remaining > 0:
            raise SlurmResourceNotEnoughException(
                f"{req.name}: {remaining} of {req.n_tasks} tasks ({req.per_task}) do not fit on the free nodes "
                f"{ {h: free[h] for h in cand} }")
        placed[req.name] = Placement(req.name, hosts, local)
    return [placed[r.name] for r in requests], free
