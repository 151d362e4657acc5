"""Synchronous PPO / GRPO on math prompts: generate -> (reward, reference log-probs, values) -> train.

All calls are MFCs of one data-flow graph executed by the master every step; answers come from the rollout
engine inside the actor's model workers. Role of ``realhf/experiments/common/ppo_math_exp.py``.
"""
import dataclasses
from typing import Any, Dict, List, Optional

from ..api.cli_args import GenerationHyperparameters, MicroBatchSpec
from ..system.dfg import MFCDef, ModelInterfaceAbstraction, ModelInterfaceType, ModelName
from ..system.system_api import (DatasetAbstraction, Experiment, ExperimentConfig, ExperimentSaveEvalControl,
                                 ExperimentScheduling, Scheduling, TasksGroup, register_experiment)
from .common import (attach_offload_hooks, resolve_allocation_mode, ModelTrainEvalConfig, parse_layout,
                     symmetric_model_workers)


@dataclasses.dataclass
class PPOHyperparameters:
    gen: GenerationHyperparameters = dataclasses.field(default_factory=GenerationHyperparameters)
    ppo_n_minibatches: int = 4
    eps_clip: float = 0.2
    c_clip: Optional[float] = None
    value_eps_clip: float = 0.2
    max_reward_clip: float = 20.0
    reward_output_scaling: float = 1.0
    reward_output_bias: float = 0.0
    kl_ctl: float = 0.0
    discount: float = 1.0
    gae_lambda: float = 1.0
    adv_norm: bool = True
    group_adv_norm: bool = False
    mask_no_eos_with_zero: bool = False
    use_decoupled_loss: bool = False
    behav_imp_weight_cap: Optional[float] = None
    recompute_logprob: bool = False
    disable_value: bool = True  # GRPO-style: no critic
    temperature: float = 1.0


def ppo_rpcs(ppo: PPOHyperparameters, group_size: int, n_seqs: int, mb_spec: MicroBatchSpec, with_gen: bool,
             with_reward_mfc: bool, reward_args: Optional[Dict[str, Any]] = None, gen_model=None) -> List[MFCDef]:
    """The MFC graph of (a)synchronous PPO; asynchronous runs drop `actor_gen` (rollout workers generate)
    and usually the reward call (the environment scored the answers already)."""
    actor_args = dict(n_minibatches=ppo.ppo_n_minibatches, generation_config=dataclasses.asdict(ppo.gen),
                      kl_ctl=ppo.kl_ctl, adv_norm=ppo.adv_norm, group_adv_norm=ppo.group_adv_norm, group_size=group_size,
                      discount=ppo.discount, gae_lambda=ppo.gae_lambda, eps_clip=ppo.eps_clip, c_clip=ppo.c_clip,
                      max_reward_clip=ppo.max_reward_clip, reward_output_scaling=ppo.reward_output_scaling,
                      reward_output_bias=ppo.reward_output_bias, mask_no_eos_with_zero=ppo.mask_no_eos_with_zero,
                      temperature=ppo.temperature, use_decoupled_loss=ppo.use_decoupled_loss,
                      behav_imp_weight_cap=ppo.behav_imp_weight_cap, recompute_logprob=ppo.recompute_logprob)
    critic_args = dict(n_minibatches=ppo.ppo_n_minibatches, kl_ctl=ppo.kl_ctl, discount=ppo.discount,
                       gae_lambda=ppo.gae_lambda, value_eps_clip=ppo.value_eps_clip,
                       max_reward_clip=ppo.max_reward_clip, reward_output_scaling=ppo.reward_output_scaling,
                       reward_output_bias=ppo.reward_output_bias, mask_no_eos_with_zero=ppo.mask_no_eos_with_zero,
                       group_size=group_size)
    actor_itf = ModelInterfaceAbstraction("ppo_actor", args=actor_args)
    rpcs: List[MFCDef] = []
    gen_keys = ("packed_input_ids", "prompt_mask", "packed_logprobs", "seq_no_eos_mask", "version_start", "version_end")
    if with_gen:
        rpcs.append(MFCDef(name="actor_gen", n_seqs=n_seqs, interface_type=ModelInterfaceType.GENERATE,
                           interface_impl=actor_itf, model_name=gen_model or "actor",
                           input_keys=("packed_prompts", "task_ids"),
                           output_keys=gen_keys, mb_spec=mb_spec))
    if with_reward_mfc:
        rpcs.append(MFCDef(name="rew_inf", n_seqs=n_seqs, interface_type=ModelInterfaceType.INFERENCE,
                           interface_impl=ModelInterfaceAbstraction("rw-math-code", args=reward_args or {}),
                           model_name="reward", input_keys=("packed_input_ids", "prompt_mask"), output_keys=("rewards",),
                           mb_spec=mb_spec))
    train_in = ["packed_input_ids", "prompt_mask", "packed_logprobs", "rewards", "seq_no_eos_mask", "version_start",
                "version_end"]
    if ppo.kl_ctl != 0:
        rpcs.append(MFCDef(name="ref_inf", n_seqs=n_seqs, interface_type=ModelInterfaceType.INFERENCE,
                           interface_impl=actor_itf, model_name="ref", input_keys=("packed_input_ids",),
                           output_keys=("packed_ref_logprobs",), output_key_remap=dict(logprobs="packed_ref_logprobs"),
                           mb_spec=mb_spec))
        train_in.append("packed_ref_logprobs")
    if ppo.use_decoupled_loss or ppo.recompute_logprob:
        rpcs.append(MFCDef(name="actor_inf", n_seqs=n_seqs, interface_type=ModelInterfaceType.INFERENCE,
                           interface_impl=actor_itf, model_name="actor", input_keys=("packed_input_ids",),
                           output_keys=("proximal_logprobs",), output_key_remap=dict(logprobs="proximal_logprobs"),
                           mb_spec=mb_spec))
        train_in.append("proximal_logprobs")
    if not ppo.disable_value:
        critic_itf = ModelInterfaceAbstraction("ppo_critic", args=critic_args)
        rpcs.append(MFCDef(name="critic_inf", n_seqs=n_seqs, interface_type=ModelInterfaceType.INFERENCE,
                           interface_impl=critic_itf, model_name="critic", input_keys=("packed_input_ids",),
                           output_keys=("values",), mb_spec=mb_spec))
        train_in.append("values")
        rpcs.append(MFCDef(name="critic_train", n_seqs=n_seqs, interface_type=ModelInterfaceType.TRAIN_STEP,
                           interface_impl=critic_itf, model_name="critic", input_keys=tuple(train_in),
                           log_return_value=True, mb_spec=mb_spec))
    rpcs.append(MFCDef(name="actor_train", n_seqs=n_seqs, interface_type=ModelInterfaceType.TRAIN_STEP,
                       interface_impl=actor_itf, model_name="actor", input_keys=tuple(train_in), log_return_value=True,
                       mb_spec=mb_spec))
    return rpcs


@dataclasses.dataclass
class PPOMATHExperiment(Experiment):
    experiment_name: str = "ppo-math"
    trial_name: str = "trial"
    actor: ModelTrainEvalConfig = dataclasses.field(default_factory=ModelTrainEvalConfig)
    ref: Optional[ModelTrainEvalConfig] = None
    critic: Optional[ModelTrainEvalConfig] = None
    ppo: PPOHyperparameters = dataclasses.field(default_factory=PPOHyperparameters)
    allocation_mode: Optional[str] = None  # e.g. "sglang.d4p1m1+d2p2m1"; overrides layout / server counts
    layout: str = "d1t1p1"
    # Layout of a second, generation-only copy of the actor over the same workers (tp = pp = 1, e.g. "d4" next to
    # a "d2t2" trainer). Its weights are re-allocated from the trainer's shards after every actor train step,
    # worker to worker, without a checkpoint. None: the trainer's own replica generates (needs tp = pp = 1).
    gen_layout: Optional[str] = None
    dataset: DatasetAbstraction = dataclasses.field(default_factory=lambda: DatasetAbstraction("math_code_prompt"))
    reward_args: Dict[str, Any] = dataclasses.field(default_factory=dict)
    train_bs_n_seqs: int = 8
    group_size: int = 1
    mb_spec: MicroBatchSpec = dataclasses.field(default_factory=MicroBatchSpec)
    exp_ctrl: ExperimentSaveEvalControl = dataclasses.field(default_factory=ExperimentSaveEvalControl)
    device: str = "cuda"
    fileroot: str = "/tmp/areal"
    seed: int = 1

    def _models(self) -> Dict[str, ModelTrainEvalConfig]:
        models = {"actor": self.actor, "reward": ModelTrainEvalConfig(backend="null")}
        if self.ppo.kl_ctl != 0:
            models["ref"] = self.ref or dataclasses.replace(self.actor, backend="inference")
        if not self.ppo.disable_value:
            models["critic"] = self.critic or dataclasses.replace(self.actor, is_critic=True, init_critic_from_actor=True)
        return models

    def scheduling_setup(self) -> ExperimentScheduling:
        resolve_allocation_mode(self)
        n = parse_layout(self.layout).world_size
        return ExperimentScheduling(master_worker=TasksGroup(1, Scheduling(cpu=4, mem=16000)),
                                    model_worker=TasksGroup(n, Scheduling(cpu=8, gpu=1, mem=64000)))

    def initial_setup(self) -> ExperimentConfig:
        resolve_allocation_mode(self)
        self.ppo.gen.n_samples = self.group_size
        replicas, gen_model = None, None
        if self.gen_layout:
            gen_model = ModelName("actor", 1)
            replicas = {gen_model: (dataclasses.replace(self.actor, backend="inference"), parse_layout(self.gen_layout))}
        rpcs = ppo_rpcs(self.ppo, self.group_size, self.train_bs_n_seqs, self.mb_spec, with_gen=True,
                        with_reward_mfc=True, reward_args=self.reward_args, gen_model=gen_model)
        attach_offload_hooks(rpcs, self._models())
        workers = symmetric_model_workers(
            self._models(), parse_layout(self.layout), self.experiment_name, self.trial_name, self.device, self.fileroot,
            f"{self.fileroot}/param_realloc/{self.experiment_name}/{self.trial_name}", self.seed, [self.dataset],
            self.actor.tokenizer_path, replicas=replicas)
        return ExperimentConfig(exp_ctrl=self.exp_ctrl, model_rpcs=rpcs, model_worker=workers, fileroot=self.fileroot)


register_experiment("ppo-math", PPOMATHExperiment)
