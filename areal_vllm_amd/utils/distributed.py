"""A second communicator world next to the default process group.

Used for trainer -> rollout-server weight transfer: trainer rank 0 and every rank of every rollout
server join one group through a TCP rendezvous with a PrefixStore, independent of each side's own
default group (``areal/utils/distributed.py:7-73`` in the reference does the same for NCCL).
"""
from datetime import timedelta
from typing import Optional, Union

import torch
import torch.distributed as dist
from torch.distributed.distributed_c10d import Backend, PrefixStore, _new_process_group_helper, _world, rendezvous

DEFAULT_TIMEOUT = timedelta(seconds=7200)


def init_custom_process_group(backend: Union[str, Backend, None] = None, init_method: Optional[str] = None,
                              timeout: Optional[timedelta] = None, world_size: int = -1, rank: int = -1,
                              store=None, group_name: Optional[str] = None, pg_options=None):
    assert (store is None) or (init_method is None), "Cannot specify both init_method and store."
    if store is not None:
        assert world_size > 0 and rank >= 0
    elif init_method is None:
        init_method = "env://"
    backend = Backend(backend) if backend else Backend("undefined")
    timeout = timeout or DEFAULT_TIMEOUT
    if store is None:
        if init_method.startswith("tcp://"):
            # Own TCP store, hosted by rank 0 of THIS group. `rendezvous()` would, inside a process started by
            # torchrun with a static rendezvous (TORCHELASTIC_USE_AGENT_STORE), make every rank - rank 0
            # included - a client of a store that nobody hosts at this address.
            host, port = init_method[len("tcp://"):].rsplit(":", 1)
            assert world_size > 0 and rank >= 0
            store = dist.TCPStore(host, int(port.split("?")[0]), world_size, rank == 0, timeout, multi_tenant=True)
        else:
            store, rank, world_size = next(iter(rendezvous(init_method, rank, world_size, timeout=timeout)))
        store.set_timeout(timeout)
        store = PrefixStore(group_name, store)
    kw = {}
    if pg_options is not None:
        kw["backend_options"] = pg_options
    pg, _ = _new_process_group_helper(world_size, rank, [], backend, store, group_name=group_name, timeout=timeout,
                                      **kw)
    _world.pg_group_ranks[pg] = {i: i for i in range(world_size)}
    return pg
