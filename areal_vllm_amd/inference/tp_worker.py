"""Tensor-parallel serving: one process per GPU of a rollout server.

The reference gets generation TP from SGLang's ``--tp-size`` (``areal/api/cli_args.py:398-527``) or from
its patched vLLM executor (``realhf/impl/model/backend/thirdparty/vllm/executor.py``). Here rank 0 of the
server's TP group is the *leader*: it owns the HTTP front end, the C++ scheduler and the block manager,
and before running a prefill / decode plan (or a weight update) it broadcasts the plan as a host object
to the *followers*, which execute the same kernels on their weight and KV-head shards. The only device
collectives per layer are the two row-parallel all-reduces over xGMI; sampling needs none because the
LM head is replicated and the sampler is keyed by (seed, step) (plus a token broadcast as a guard).

``python -m areal_vllm_amd.inference.server --tp-size N ...`` lands in ``launch``: the first process picks
a rendezvous port, starts N-1 copies of itself with ``AREAL_TP_RANK`` set and becomes rank 0.
"""
import atexit
import os
import signal
import subprocess
import sys
from typing import List, Optional

import torch
import torch.distributed as dist

from ..api.cli_args import InferenceServerConfig
from ..parallel.grid import DIST_TIMEOUT
from ..utils import logging, network
from .engine import NativeInferenceEngine

logger = logging.getLogger("tp-worker")


def build_tp_engine(args, cfg: InferenceServerConfig, rank: int, init_addr: str,
                    initialize: bool = True) -> NativeInferenceEngine:
    """Join the TP group of this server (RCCL for tensors, gloo for host objects) and build rank `rank`'s
    engine shard."""
    tp = int(args.tp_size)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if init_addr.split(":")[0] in ("127.0.0.1", "localhost"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    use_gpu = torch.cuda.is_available() and (getattr(args, "device", None) or "cuda").startswith("cuda")
    if use_gpu:
        dev = torch.device("cuda", (int(args.base_gpu_id) + rank) % torch.cuda.device_count())
        torch.cuda.set_device(dev)
        # AREAL_TP_BACKEND=gloo: rehearsal of a server's ranks on ONE GPU (RCCL cannot host two ranks on a device)
        backend = os.environ.get("AREAL_TP_BACKEND") or "cpu:gloo,cuda:nccl"
    else:
        dev = torch.device("cpu")
        backend = "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method=f"tcp://{init_addr}", rank=rank, world_size=tp,
                                timeout=DIST_TIMEOUT)
    eng = NativeInferenceEngine(cfg, device=dev, tp_size=tp, tp_rank=rank, tp_group=dist.group.WORLD,
                                num_pages=getattr(args, "num_pages", None),
                                kv_cache_gb=getattr(args, "kv_cache_gb", None))
    if initialize:
        eng.initialize(load_weights=True)
    return eng


def launch(args, cfg: InferenceServerConfig, argv: List[str]) -> Optional[NativeInferenceEngine]:
    """Returns the leader's engine, or None in a follower process after it has been shut down."""
    rank = int(os.environ.get("AREAL_TP_RANK", "-1"))
    init_addr = os.environ.get("AREAL_TP_INIT")
    if rank < 0:
        init_addr = args.dist_init_addr or f"127.0.0.1:{network.find_free_ports(1)[0]}"
        procs = []
        for r in range(1, int(args.tp_size)):
            env = dict(os.environ, AREAL_TP_RANK=str(r), AREAL_TP_INIT=init_addr)
            procs.append(subprocess.Popen([sys.executable, "-m", "areal_vllm_amd.inference.server", *argv], env=env))

        def _reap():
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()

        atexit.register(_reap)
        # SIGTERM (how launchers stop a server) must run the atexit hooks, or the followers would linger
        signal.signal(signal.SIGTERM, lambda *_: sys.exit(0))
        rank = 0
    eng = build_tp_engine(args, cfg, rank, init_addr)
    if rank > 0:
        eng.run_follower()
        dist.destroy_process_group()
        return None
    logger.info(f"tensor-parallel server: {args.tp_size} ranks, rendezvous {init_addr}")
    return eng
