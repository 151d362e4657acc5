"""Per-node rollout-server supervisor used by the multi-node launchers.

One wrapper process runs per node (or per GPU slice of a node): it derives how many native rollout
servers fit on the GPUs it can see, starts them with disjoint port ranges and per-replica seeds, waits
for ``/health``, registers ``host:port`` under ``gen_servers`` in name_resolve and then watches the
children - when one exits, the rest are terminated and the wrapper exits non-zero so that the
launcher's recover logic sees a failed job. Behaviour of ``areal/launcher/sglang_server.py:115-254``;
multi-node replicas use ``AREAL_SERVER_MULTI_NODE_{RANK,MASTER_ADDR,MASTER_PORT}``.

Run: ``python -m areal_vllm_amd.launcher.server_wrapper --config <yaml> [k=v ...]``
"""
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional

import requests

from ..api.alloc_mode import AllocationMode
from ..api.cli_args import InferenceServerConfig
from ..utils import logging, name_resolve, names
from ..utils.launcher import visible_devices
from ..utils.network import find_free_ports, gethostip

logger = logging.getLogger("server-wrapper")


def wait_for_server(base_url: str, proc: Optional[subprocess.Popen] = None, timeout: Optional[float] = None):
    start = time.time()
    while True:
        if proc is not None and proc.poll() is not None:
            raise RuntimeError(f"rollout server exited with code {proc.returncode} before becoming healthy")
        try:
            if requests.get(f"{base_url}/health", timeout=10).status_code == 200:
                return
        except requests.exceptions.RequestException:
            pass
        if timeout and time.time() - start > timeout:
            raise TimeoutError("Server did not become ready within timeout period")
        time.sleep(1)


class ServerWrapper:
    def __init__(self, experiment_name: str, trial_name: str, server_config: InferenceServerConfig,
                 allocation_mode: AllocationMode, n_gpus_per_node: int):
        self.experiment_name, self.trial_name = experiment_name, trial_name
        self.config = server_config
        self.allocation_mode = allocation_mode
        self.n_gpus_per_node = n_gpus_per_node
        self.processes: List[subprocess.Popen] = []

    def plan(self):
        """-> list of (cmd, host, port, node_rank) for the servers this process must start."""
        gen = self.allocation_mode.gen
        gpus_per_server = gen.tp_size * gen.pp_size
        cross_nodes = gpus_per_server > self.n_gpus_per_node
        if cross_nodes:
            assert gpus_per_server % self.n_gpus_per_node == 0, "a multi-node replica must use whole nodes"
            node_rank = int(os.environ["AREAL_SERVER_MULTI_NODE_RANK"])
            master = f"{os.environ['AREAL_SERVER_MULTI_NODE_MASTER_ADDR']}:{os.environ['AREAL_SERVER_MULTI_NODE_MASTER_PORT']}"
        else:
            node_rank, master = 0, None
        n_per_node = max(1, self.n_gpus_per_node // gpus_per_server)
        devs = visible_devices()
        if devs and len(devs) < self.n_gpus_per_node:
            n_here = max(1, len(devs) // gpus_per_server)
            offset = min(int(d) for d in devs) // gpus_per_server
        else:
            n_here, offset = n_per_node, 0
        ports_per_server = 40000 // n_per_node
        host = gethostip()
        base_seed = self.config.random_seed
        out = []
        for idx in range(offset, offset + n_here):
            lo = 10000 + idx * ports_per_server
            port, dist_port = find_free_ports(2, (lo, lo + ports_per_server))
            self.config.random_seed = base_seed + idx
            cmd = InferenceServerConfig.build_cmd(
                self.config, tp_size=gen.tp_size, base_gpu_id=(idx - offset) * gpus_per_server, host=host, port=port,
                dist_init_addr=master if cross_nodes else f"127.0.0.1:{dist_port}",
                n_nodes=max(1, gpus_per_server // self.n_gpus_per_node), node_rank=node_rank)
            out.append((cmd, host, port, node_rank))
        self.config.random_seed = base_seed
        return out

    def register_when_healthy(self, p: subprocess.Popen, host: str, port: int, node_rank: int):
        wait_for_server(f"http://{host}:{port}", p)
        if node_rank == 0:
            name_resolve.add_subentry(names.gen_servers(self.experiment_name, self.trial_name), f"{host}:{port}")
        logger.info(f"rollout server launched at http://{host}:{port}")

    def run(self) -> int:
        """Start the servers and watch them until one exits or this process is told to stop. The servers run in
        sessions of their own, so nothing reaches them by itself when the wrapper's job is cancelled (`scancel`,
        a cancelled Ray task, the local launcher's SIGTERM): every way out of here stops them."""
        import signal
        import threading

        def on_signal(signum, _frame):
            raise KeyboardInterrupt(f"signal {signum}")

        if threading.current_thread() is threading.main_thread():
            for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
                signal.signal(sig, on_signal)
        plan = self.plan()
        try:
            for a in plan:  # started one after another here so that an interrupt finds every process in the list
                self.processes.append(subprocess.Popen(a[0].split(), start_new_session=True))
            with ThreadPoolExecutor(max_workers=max(1, len(plan))) as ex:
                list(ex.map(lambda ap: self.register_when_healthy(ap[1], *ap[0][1:]), zip(plan, self.processes)))
            while True:
                for i, p in enumerate(self.processes):
                    rc = p.poll()
                    if rc is not None:
                        logger.error(f"rollout server {plan[i][1]}:{plan[i][2]} exited with code {rc}; "
                                     "stopping the others")
                        return rc if rc != 0 else 1
                time.sleep(1)
        except KeyboardInterrupt as e:
            logger.info(f"server wrapper interrupted ({e}); stopping {len(self.processes)} rollout server(s)")
            return 0
        finally:
            self.stop()

    def stop(self):
        for p in self.processes:
            if p.poll() is None:
                p.terminate()
        for p in self.processes:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()


def main(argv: Optional[List[str]] = None):
    from .local import load_launcher_config

    config, _raw, _ = load_launcher_config(sys.argv[1:] if argv is None else argv)
    name_resolve.reconfigure(config.cluster.name_resolve)
    am = AllocationMode.from_str(config.allocation_mode)
    w = ServerWrapper(config.experiment_name, config.trial_name, config.sglang, am, config.cluster.n_gpus_per_node)
    sys.exit(w.run())


if __name__ == "__main__":
    main()
