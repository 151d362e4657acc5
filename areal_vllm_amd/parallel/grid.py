"""Rank grid and process groups for (pp, dp, cp, tp) parallelism.

Rank order is tp fastest, then cp (Ulysses sequence parallel), then dp, then pp - the
"tp-cp-ep-dp-pp" order the reference passes to Megatron (``areal/experimental/megatron_engine.py:158-179``)
and the same shape as its FSDP device mesh (dp, sp, tp) (``areal/engine/fsdp_engine.py:114-165``).
On one MI355X node every group has at most 8 ranks, so all collectives stay on xGMI; every group is
created up-front (RCCL communicators are not created lazily on the hot path).
"""
import dataclasses
import datetime
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from ..api.alloc_mode import ParallelStrategy

DIST_TIMEOUT = datetime.timedelta(seconds=7200)


@dataclasses.dataclass
class GridCoord:
    pp: int
    dp: int
    cp: int
    tp: int


class ProcessGrid:
    def __init__(self, strategy: ParallelStrategy, rank_offset: int = 0, ranks: Optional[List[int]] = None,
                 backend: Optional[str] = None):
        self.strategy = strategy
        self.tp, self.cp, self.dp, self.pp = strategy.tp_size, strategy.cp_size, strategy.dp_size, strategy.pp_size
        self.ep = strategy.ep_size
        # expert tensor parallel size: the ranks one expert is split over - the tp group (default), or 1: whole
        # experts, and the tp ranks join the expert-parallel / expert-data-parallel dimensions instead
        self.etp = int(getattr(strategy, "etp_size", None) or self.tp) if self.ep > 1 else self.tp
        self.vpp = max(1, int(getattr(strategy, "virtual_pipeline_parallel_size", None) or 1))
        self.links: Dict[str, Optional[dist.ProcessGroup]] = {}
        self.world = self.tp * self.cp * self.dp * self.pp
        self.ranks = ranks if ranks is not None else list(range(rank_offset, rank_offset + self.world))
        assert len(self.ranks) == self.world
        self.global_rank = dist.get_rank() if dist.is_initialized() else 0
        self.member = self.global_rank in self.ranks
        self.local_index = self.ranks.index(self.global_rank) if self.member else -1
        self.coord = self.coord_of(self.local_index) if self.member else None
        self.backend = backend
        self.groups: Dict[str, Optional[dist.ProcessGroup]] = {}
        self._group_cache: Dict[tuple, dist.ProcessGroup] = {}
        self.group_ranks: Dict[str, List[int]] = {}
        if dist.is_initialized():
            self._build_groups()
        else:
            for k in ("tp", "cp", "dp", "pp", "mp", "dp_cp", "world", "ep", "edp"):
                self.groups[k] = None
                self.group_ranks[k] = [self.global_rank]

    def coord_of(self, idx: int) -> GridCoord:
        tp = idx % self.tp
        cp = (idx // self.tp) % self.cp
        dp = (idx // (self.tp * self.cp)) % self.dp
        pp = idx // (self.tp * self.cp * self.dp)
        return GridCoord(pp, dp, cp, tp)

    def index_of(self, pp: int, dp: int, cp: int, tp: int) -> int:
        return ((pp * self.dp + dp) * self.cp + cp) * self.tp + tp

    def _new_group(self, name: str, rank_lists: List[List[int]]):
        """Groups are created collectively (every process walks every rank list) and shared between
        dimensions with identical members: with tp = cp = pp = 1 the dp, dp_cp and world groups are one
        communicator and the four kinds of single-rank groups are one object per rank, which keeps
        start-up at a handful of communicator creations instead of dozens."""
        mine, mine_ranks = None, None
        for rl in rank_lists:
            key = tuple(rl)
            g = self._group_cache.get(key)
            if g is None:
                g = dist.new_group(ranks=rl, timeout=DIST_TIMEOUT, backend=self.backend)
                self._group_cache[key] = g
            if self.global_rank in rl:
                mine, mine_ranks = g, rl
        self.groups[name] = mine
        self.group_ranks[name] = mine_ranks or []

    def _build_groups(self):
        R = lambda pp, dp, cp, tp: self.ranks[self.index_of(pp, dp, cp, tp)]  # noqa: E731
        rng = range
        self._new_group("tp", [[R(p, d, c, t) for t in rng(self.tp)]
                               for p in rng(self.pp) for d in rng(self.dp) for c in rng(self.cp)])
        self._new_group("cp", [[R(p, d, c, t) for c in rng(self.cp)]
                               for p in rng(self.pp) for d in rng(self.dp) for t in rng(self.tp)])
        self._new_group("dp", [[R(p, d, c, t) for d in rng(self.dp)]
                               for p in rng(self.pp) for c in rng(self.cp) for t in rng(self.tp)])
        self._new_group("pp", [[R(p, d, c, t) for p in rng(self.pp)]
                               for d in rng(self.dp) for c in rng(self.cp) for t in rng(self.tp)])
        # model-parallel group of one data-parallel replica: pp x cp x tp (receives the rollout batch)
        self._new_group("mp", [[R(p, d, c, t) for p in rng(self.pp) for c in rng(self.cp) for t in rng(self.tp)]
                               for d in rng(self.dp)])
        # gradient-reduction group: parameters are replicated over dp x cp
        self._new_group("dp_cp", [[R(p, d, c, t) for d in rng(self.dp) for c in rng(self.cp)]
                                  for p in rng(self.pp) for t in rng(self.tp)])
        self._new_group("world", [list(self.ranks)])
        if self.pp > 1:
            # first + last stage of every pipeline: owners of tied input / output embeddings
            self._new_group("embed", [[R(0, d, c, t), R(self.pp - 1, d, c, t)]
                                      for d in rng(self.dp) for c in rng(self.cp) for t in rng(self.tp)])
        if self.pp > 1 and self.vpp > 1:
            self._build_pipeline_links()
        if self.ep > 1 and self.etp != self.tp:
            # Whole experts: per pipeline stage the dp x tp ranks (tp fastest, the order of the rank grid) are
            # cut into consecutive expert-parallel groups of `ep` ranks; ranks at the same position of different
            # groups hold the same experts (expert data parallel). Order "etp-ep-edp-pp" of the reference's
            # Megatron set-up (``areal/experimental/megatron_engine.py:158-179``) with etp = 1.
            assert self.etp == 1 and self.cp == 1, "expert tensor parallel size must be the tp size or 1 (and cp 1)"
            S = self.dp * self.tp
            assert S % self.ep == 0 and (self.ep % self.tp == 0 or self.tp % self.ep == 0), \
                "with whole experts the expert parallel size must divide dp x tp and be a multiple or a divisor of tp"
            Rs = lambda p, s: R(p, s // self.tp, 0, s % self.tp)  # noqa: E731
            self._new_group("ep", [[Rs(p, b * self.ep + e) for e in rng(self.ep)]
                                   for p in rng(self.pp) for b in rng(S // self.ep)])
            self._new_group("edp", [[Rs(p, b * self.ep + e) for b in rng(S // self.ep)]
                                    for p in rng(self.pp) for e in rng(self.ep)])
        elif self.ep > 1:
            assert self.dp % self.ep == 0, "expert parallel size must divide the data parallel size"
            edp = self.dp // self.ep
            self._new_group("ep", [[R(p, e0 * self.ep + e, c, t) for e in rng(self.ep)]
                                   for p in rng(self.pp) for e0 in rng(edp) for c in rng(self.cp)
                                   for t in rng(self.tp)])
            self._new_group("edp", [[R(p, e0 * self.ep + e, c, t) for e0 in rng(edp) for c in rng(self.cp)]
                                    for p in rng(self.pp) for e in rng(self.ep) for t in rng(self.tp)])
        else:
            self.groups["ep"], self.group_ranks["ep"] = None, [self.global_rank]
            self.groups["edp"], self.group_ranks["edp"] = self.groups["dp_cp"], self.group_ranks["dp_cp"]

    def _build_pipeline_links(self):
        """Interleaved schedules send activations round the ring of stages (last -> first between two layer
        chunks) with sends and receives of several micro-batches in flight. Every directed neighbour pair
        gets a two-rank communicator per direction of travel, so a transfer on one link never queues behind a
        transfer on another: `send_fwd` (me -> next), `recv_fwd` (prev -> me), `send_bwd` (me -> prev),
        `recv_bwd` (next -> me)."""
        R = lambda pp, dp, cp, tp: self.ranks[self.index_of(pp, dp, cp, tp)]  # noqa: E731
        for d in range(self.dp):
            for c in range(self.cp):
                for t in range(self.tp):
                    fwd, bwd = [], []
                    for p in range(self.pp):
                        nxt, prv = (p + 1) % self.pp, (p - 1) % self.pp
                        fwd.append(dist.new_group(ranks=[R(p, d, c, t), R(nxt, d, c, t)], timeout=DIST_TIMEOUT,
                                                  backend=self.backend))
                        bwd.append(dist.new_group(ranks=[R(p, d, c, t), R(prv, d, c, t)], timeout=DIST_TIMEOUT,
                                                  backend=self.backend))
                    me = self.coord
                    if me is not None and (me.dp, me.cp, me.tp) == (d, c, t):
                        p = me.pp
                        self.links = dict(send_fwd=fwd[p], recv_fwd=fwd[(p - 1) % self.pp], send_bwd=bwd[p],
                                          recv_bwd=bwd[(p + 1) % self.pp])
                        self.ring_next = R((p + 1) % self.pp, d, c, t)
                        self.ring_prev = R((p - 1) % self.pp, d, c, t)

    # ---------------------------------------------------------------- convenience accessors
    def group(self, name: str):
        return self.groups.get(name)

    def size(self, name: str) -> int:
        return {"tp": self.tp, "cp": self.cp, "dp": self.dp, "pp": self.pp, "mp": self.pp * self.cp * self.tp,
                "dp_cp": self.dp * self.cp, "world": self.world, "ep": self.ep}[name]

    def rank_in(self, name: str) -> int:
        c = self.coord
        if c is None:
            return -1
        return {"tp": c.tp, "cp": c.cp, "dp": c.dp, "pp": c.pp, "dp_cp": c.dp * self.cp + c.cp,
                "world": self.local_index, "ep": self.expert_slot() % max(1, self.ep),
                "mp": (c.pp * self.cp + c.cp) * self.tp + c.tp}[name]

    def expert_slot(self) -> int:
        """Position of this rank in the expert grid of its pipeline stage: `slot % ep` is its expert-parallel rank
        (which experts it holds), `slot // ep` its expert-data-parallel replica (cp aside)."""
        c = self.coord
        return c.dp * self.tp + c.tp if self.etp != self.tp else c.dp

    def holds_first_expert_replica(self) -> bool:
        """One rank per distinct expert shard: the checkpoint writers of the routed experts."""
        return self.coord.cp == 0 and self.expert_slot() < self.ep

    def is_dp_head(self) -> bool:
        c = self.coord
        return c is not None and c.tp == 0 and c.cp == 0 and c.pp == self.pp - 1

    def dp_head_global_rank(self) -> int:
        c = self.coord
        return self.ranks[self.index_of(self.pp - 1, c.dp, 0, 0)]

    def is_first_stage(self) -> bool:
        return self.coord.pp == 0

    def is_last_stage(self) -> bool:
        return self.coord.pp == self.pp - 1

    def prev_stage_rank(self) -> int:
        c = self.coord
        return self.ranks[self.index_of(c.pp - 1, c.dp, c.cp, c.tp)]

    def next_stage_rank(self) -> int:
        c = self.coord
        return self.ranks[self.index_of(c.pp + 1, c.dp, c.cp, c.tp)]


def init_distributed(backend: Optional[str] = None, timeout=DIST_TIMEOUT):
    """Initialise torch.distributed from the torchrun environment (RCCL on GPU, gloo on CPU)."""
    if dist.is_initialized():
        return
    from ..platforms import current_platform

    backend = backend or current_platform.communication_backend
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
        # single-node run: keep gloo off hostname resolution (container hostnames may not resolve)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from ..utils import host_staged

    if host_staged.maybe_install_from_env():  # one-GPU rehearsal of a multi-rank layout: host backend, staged copies
        backend = "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    if backend == "nccl":
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank % max(1, torch.cuda.device_count()))))
        torch.cuda.set_device(local_rank)
        # gloo is registered next to RCCL so host-side object collectives never touch the GPU. No
        # `device_id`: communicators are then created by plain ncclCommInitRank on first use of each
        # group instead of eagerly by splitting the world communicator.
        dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world, timeout=timeout)
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
