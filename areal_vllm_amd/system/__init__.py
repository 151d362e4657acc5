"""Worker-based runtime: a master worker drives a data-flow graph of model function calls on model
workers, rollout workers stream trajectories from generation servers through a routing manager.

This is the multi-process control plane of the reference's ``realhf/system`` re-designed around plain
TCP sockets + name_resolve (no ZeroMQ dependency) and asyncio workers; the data plane stays on
torch.distributed (RCCL over xGMI on MI355X, gloo in CPU tests).
"""
