"""Generation-server worker: owns one native rollout server subprocess.

Builds the server command from the experiment's server config, starts it on the GPUs assigned to this
worker, waits for ``/health``, registers ``host:port`` under ``gen_servers`` and afterwards only watches
the child: a dead server makes the worker fail so that the controller's recover logic kicks in.
Behaviour of ``realhf/system/generation_server.py:90-200`` (which spawns SGLang).
"""
import dataclasses
import os
import subprocess
import time
from typing import List, Optional

import requests

from ..api.cli_args import InferenceServerConfig
from ..utils import logging, name_resolve, names, network
from ..utils.launcher import device_env
from .worker_base import PollResult, Worker

logger = logging.getLogger("generation-server")


@dataclasses.dataclass
class GenerationServerConfig:
    server: InferenceServerConfig = dataclasses.field(default_factory=InferenceServerConfig)
    tp_size: int = 1
    base_gpu_id: int = 0
    device_ids: Optional[List[int]] = None
    extra_args: List[str] = dataclasses.field(default_factory=list)
    startup_timeout: float = 600.0


class GenerationServer(Worker):
    def _configure(self, config: GenerationServerConfig):
        self.config = config
        self.proc: Optional[subprocess.Popen] = None
        self.addr: Optional[str] = None
        return self._info

    def launch_server_subprocess(self):
        cfg = self.config
        host = network.gethostip()
        port, dist_port = network.find_free_ports(2)
        cfg.server.random_seed = cfg.server.random_seed + self.worker_index
        cmd = InferenceServerConfig.build_cmd(cfg.server, tp_size=cfg.tp_size, base_gpu_id=cfg.base_gpu_id, host=host,
                                              port=port, dist_init_addr=f"127.0.0.1:{dist_port}")
        env = dict(os.environ)
        if cfg.device_ids is not None:
            env.update(device_env([str(d) for d in cfg.device_ids]))
        self.proc = subprocess.Popen(cmd.split() + list(cfg.extra_args), env=env, start_new_session=True)
        self.addr = f"{host}:{port}"
        deadline = time.time() + cfg.startup_timeout
        while time.time() < deadline:
            if self.proc.poll() is not None:
                raise RuntimeError(f"rollout server exited with code {self.proc.returncode} during start-up")
            try:
                if requests.get(f"http://{self.addr}/health", timeout=5).status_code == 200:
                    break
            except requests.exceptions.RequestException:
                time.sleep(0.5)
        else:
            raise TimeoutError("rollout server did not become healthy")
        name_resolve.add_subentry(names.gen_servers(self.experiment_name, self.trial_name), self.addr)
        logger.info(f"rollout server {self.worker_index} ready at {self.addr}")

    def _poll(self) -> PollResult:
        if self.proc is None:
            self.launch_server_subprocess()
        if self.proc.poll() is not None:
            raise RuntimeError(f"rollout server at {self.addr} died with code {self.proc.returncode}")
        time.sleep(1.0)
        return PollResult(0, 0)

    def _exit_hook(self, exit_status):
        if self.proc is not None and self.proc.poll() is None:
            self.proc.terminate()
            try:
                self.proc.wait(timeout=20)
            except subprocess.TimeoutExpired:
                self.proc.kill()
