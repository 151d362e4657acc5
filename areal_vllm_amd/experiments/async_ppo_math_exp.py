"""Asynchronous PPO / GRPO on math prompts: rollout workers + generation servers produce trajectories
continuously; the trainer consumes them with bounded staleness and publishes new weights after every step.

Process layout: 1 master, N model workers (actor [+ ref, critic], symmetric), G generation servers, one
generation-server manager, R rollout workers. The trainer's graph has no generation call: its source is
the stream of scored trajectories. Role of ``realhf/experiments/async_exp/async_ppo_math_exp.py`` and
``async_rl_exp.py`` (358 lines).
"""
import dataclasses
from typing import Any, Dict, List, Optional

from ..api.cli_args import InferenceServerConfig, MicroBatchSpec
from ..system.agent_api import AgentAbstraction, EnvServiceAbstraction
from ..system.generation_server import GenerationServerConfig
from ..system.gserver_manager import GserverManagerConfig
from ..system.system_api import (DatasetAbstraction, Experiment, ExperimentConfig, ExperimentSaveEvalControl,
                                 ExperimentScheduling, RolloutWorkerConfig, Scheduling, TasksGroup,
                                 register_experiment)
from .common import (attach_offload_hooks, resolve_allocation_mode, ModelTrainEvalConfig, parse_layout,
                     symmetric_model_workers)
from .ppo_math_exp import PPOHyperparameters, ppo_rpcs


@dataclasses.dataclass
class AsyncPPOMATHExperiment(Experiment):
    experiment_name: str = "async-ppo-math"
    trial_name: str = "trial"
    actor: ModelTrainEvalConfig = dataclasses.field(default_factory=ModelTrainEvalConfig)
    ref: Optional[ModelTrainEvalConfig] = None
    critic: Optional[ModelTrainEvalConfig] = None
    ppo: PPOHyperparameters = dataclasses.field(default_factory=PPOHyperparameters)
    allocation_mode: Optional[str] = None  # e.g. "sglang.d4p1m1+d2p2m1"; overrides layout / server counts
    layout: str = "d1t1p1"  # trainer layout; generation servers are data-parallel replicas with tp `gen_tp_size`
    n_generation_servers: int = 1
    gen_tp_size: int = 1
    n_rollout_workers: int = 1
    dataset: DatasetAbstraction = dataclasses.field(default_factory=lambda: DatasetAbstraction("math_code_prompt"))
    dataset_size: int = 0  # number of prompts the rollout workers iterate per epoch (epoch accounting)
    agent: AgentAbstraction = dataclasses.field(default_factory=lambda: AgentAbstraction("math-single-step"))
    env: EnvServiceAbstraction = dataclasses.field(default_factory=lambda: EnvServiceAbstraction("math-code-single-step"))
    server: InferenceServerConfig = dataclasses.field(default_factory=InferenceServerConfig)
    server_extra_args: List[str] = dataclasses.field(default_factory=list)
    train_bs_n_seqs: int = 8
    group_size: int = 1
    max_head_offpolicyness: int = 0
    max_concurrent_rollouts: int = 128
    new_tokens_per_chunk: int = 1 << 30
    schedule_policy: str = "round_robin"
    flush_request_timeout: float = 300.0
    mb_spec: MicroBatchSpec = dataclasses.field(default_factory=MicroBatchSpec)
    exp_ctrl: ExperimentSaveEvalControl = dataclasses.field(default_factory=ExperimentSaveEvalControl)
    device: str = "cuda"
    fileroot: str = "/tmp/areal"
    seed: int = 1

    @property
    def realloc_root(self) -> str:
        return f"{self.fileroot}/param_realloc/{self.experiment_name}/{self.trial_name}"

    def _models(self) -> Dict[str, ModelTrainEvalConfig]:
        models = {"actor": self.actor}
        if self.ppo.kl_ctl != 0:
            models["ref"] = self.ref or dataclasses.replace(self.actor, backend="inference")
        if not self.ppo.disable_value:
            models["critic"] = self.critic or dataclasses.replace(self.actor, is_critic=True, init_critic_from_actor=True)
        return models

    def scheduling_setup(self) -> ExperimentScheduling:
        resolve_allocation_mode(self)
        n = parse_layout(self.layout).world_size
        return ExperimentScheduling(
            master_worker=TasksGroup(1, Scheduling(cpu=4, mem=16000)),
            model_worker=TasksGroup(n, Scheduling(cpu=8, gpu=1, mem=64000)),
            generation_server=TasksGroup(self.n_generation_servers, Scheduling(cpu=8, gpu=self.gen_tp_size, mem=64000)),
            gserver_manager=TasksGroup(1, Scheduling(cpu=2, mem=4000)),
            rollout_worker=TasksGroup(self.n_rollout_workers, Scheduling(cpu=4, mem=8000)))

    def initial_setup(self) -> ExperimentConfig:
        resolve_allocation_mode(self)
        self.ppo.gen.n_samples = self.group_size
        rpcs = ppo_rpcs(self.ppo, self.group_size, self.train_bs_n_seqs, self.mb_spec, with_gen=False,
                        with_reward_mfc=False)
        attach_offload_hooks(rpcs, self._models())
        stream = DatasetAbstraction("puller_stream", args=dict(dataset_size=self.dataset_size or self.train_bs_n_seqs))
        workers = symmetric_model_workers(
            self._models(), parse_layout(self.layout), self.experiment_name, self.trial_name, self.device, self.fileroot,
            self.realloc_root, self.seed, [stream], self.actor.tokenizer_path)
        server = dataclasses.replace(self.server, model_path=self.server.model_path or self.actor.path,
                                     dtype=self.actor.dtype if self.device != "cpu" else "float32")
        extra = list(self.server_extra_args)
        if self.device == "cpu" and "--device" not in extra:
            extra += ["--device", "cpu", "--num-pages", "1024"]
        gservers = [GenerationServerConfig(server=dataclasses.replace(server), tp_size=self.gen_tp_size, extra_args=extra)
                    for _ in range(self.n_generation_servers)]
        manager = GserverManagerConfig(
            model_name="actor", n_servers=self.n_generation_servers, schedule_policy=self.schedule_policy,
            max_head_offpolicyness=self.max_head_offpolicyness, train_batch_size=self.train_bs_n_seqs,
            max_concurrent_rollouts=self.max_concurrent_rollouts, flush_request_timeout=self.flush_request_timeout,
            realloc_root=self.realloc_root, poll_interval=0.2)
        agent_args = dict(self.agent.args)
        agent_args.setdefault("gconfig", self.ppo.gen)
        agent_args.setdefault("tokenizer_path", self.actor.tokenizer_path)
        stops = list(self.ppo.gen.stop_token_ids or [])
        rollouts = [RolloutWorkerConfig(
            base_seed=self.seed, model_name="actor", tokenizer_path=self.actor.tokenizer_path,
            new_tokens_per_chunk=self.new_tokens_per_chunk, env=self.env,
            agent=AgentAbstraction(self.agent.type_, agent_args), datasets=[self.dataset], stop_token_ids=stops,
            fileroot=self.fileroot) for _ in range(self.n_rollout_workers)]
        return ExperimentConfig(exp_ctrl=self.exp_ctrl, model_rpcs=rpcs, model_worker=workers,
                                generation_server=gservers, gserver_manager=[manager], rollout_worker=rollouts,
                                fileroot=self.fileroot)


register_experiment("async-ppo-math", AsyncPPOMATHExperiment)
