"""Who owns which piece of data, and how to get it to the workers that need it next.

The master only ever sees metadata. `GlobalStorageTracker` records, per model worker, the (sample id, key)
pairs it stores; `RedistribPlanner.derive_plan` turns "these workers need these ids with these keys" into
a list of point-to-point steps (source worker, destination workers, key, ids). Role of
``realhf/system/redistributor.py`` (GlobalStorageTracker / RedistribPlanner).

MI355X-first difference: the reference derives gather / scatter / broadcast steps and runs each one as an
NCCL collective on a process group created per step shape. On one xGMI node every pair of GPUs has its
own link, so the natural primitive is a batch of direct sends: all steps of a plan are posted together
(``dist.batch_isend_irecv`` = one RCCL group call) and every link carries only the bytes its pair needs.
"""
import dataclasses
from collections import defaultdict
from typing import Dict, Hashable, List, Sequence, Set, Tuple


class GlobalStorageTracker:
    def __init__(self, world_size: int):
        self.world_size = world_size
        # worker -> id -> keys
        self.storages: List[Dict[Hashable, Set[str]]] = [defaultdict(set) for _ in range(world_size)]

    def add_data(self, worker: int, ids: Sequence[Hashable], keys: Sequence[str]):
        for i in ids:
            self.storages[worker][i].update(keys)

    def add_data_synced(self, worker: int, ids, keys):
        self.add_data(worker, ids, keys)

    def owners(self, id_: Hashable, key: str) -> List[int]:
        return [w for w in range(self.world_size) if key in self.storages[w].get(id_, ())]

    def clear_data(self, ids: Sequence[Hashable]):
        for st in self.storages:
            for i in ids:
                st.pop(i, None)

    def n_items(self) -> int:
        return len({i for st in self.storages for i in st})


@dataclasses.dataclass
class RedistribStep:
    src: int
    dsts: List[int]
    key: str
    ids: List[Hashable]

    def involves(self, worker: int) -> bool:
        return worker == self.src or worker in self.dsts


class RedistribPlanner:
    def __init__(self, tracker: GlobalStorageTracker):
        self.tracker = tracker

    def derive_plan(self, dests: Dict[int, List[Hashable]], keys: Sequence[str]) -> List[RedistribStep]:
        """`dests`: worker -> ids it must hold (with all `keys`) for the next call. Steps are emitted in a
        deterministic order so every worker walks the same list; a worker that already holds a piece gets
        nothing sent, and pieces with several owners are taken from the least-loaded one."""
        load = defaultdict(int)
        by_pair: Dict[Tuple[int, str, Tuple], List[int]] = {}
        pieces: Dict[Tuple[int, int, str], List[Hashable]] = defaultdict(list)
        for key in sorted(keys):
            for dst in sorted(dests):
                for i in dests[dst]:
                    own = self.tracker.owners(i, key)
                    if not own:
                        raise RuntimeError(f"nobody owns key '{key}' of sample {i}")
                    if dst in own:
                        continue
                    src = min(own, key=lambda w: (load[w], w))
                    load[src] += 1
                    pieces[(src, dst, key)].append(i)
        # merge destinations that need exactly the same ids of a key from the same source
        for (src, dst, key), ids in sorted(pieces.items(), key=lambda kv: (kv[0][2], kv[0][0], kv[0][1])):
            by_pair.setdefault((src, key, tuple(ids)), []).append(dst)
        steps = [RedistribStep(src=src, dsts=sorted(dsts), key=key, ids=list(ids))
                 for (src, key, ids), dsts in by_pair.items()]
        steps.sort(key=lambda s: (s.key, s.src, s.dsts, [str(i) for i in s.ids]))
        return steps

    def apply(self, plan: List[RedistribStep]):
        """Record the effect of a finished plan: destinations now own what they received."""
        for s in plan:
            for d in s.dsts:
                self.tracker.add_data(d, s.ids, [s.key])
