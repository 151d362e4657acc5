"""Offline evaluation as a worker-runtime experiment: generate answers for every prompt of a dataset and
score them with the multi-task reward interface; no training call, one pass over the data. Role of
``realhf/experiments/common/math_code_eval_exp.py`` (``math-code-eval``)."""
import dataclasses
from typing import Any, Dict

from ..api.cli_args import GenerationHyperparameters, MicroBatchSpec
from ..system.dfg import MFCDef, ModelInterfaceAbstraction, ModelInterfaceType
from ..system.system_api import (DatasetAbstraction, Experiment, ExperimentConfig, ExperimentSaveEvalControl,
                                 ExperimentScheduling, Scheduling, TasksGroup, register_experiment)
from .common import ModelTrainEvalConfig, parse_layout, symmetric_model_workers


@dataclasses.dataclass
class MathCodeEvalExperiment(Experiment):
    experiment_name: str = "math-code-eval"
    trial_name: str = "trial"
    actor: ModelTrainEvalConfig = dataclasses.field(default_factory=lambda: ModelTrainEvalConfig(backend="inference"))
    gen: GenerationHyperparameters = dataclasses.field(default_factory=GenerationHyperparameters)
    layout: str = "d1t1p1"
    dataset: DatasetAbstraction = dataclasses.field(default_factory=lambda: DatasetAbstraction("math_code_prompt"))
    reward_args: Dict[str, Any] = dataclasses.field(default_factory=dict)
    eval_bs_n_seqs: int = 8
    group_size: int = 1
    mb_spec: MicroBatchSpec = dataclasses.field(default_factory=MicroBatchSpec)
    device: str = "cuda"
    fileroot: str = "/tmp/areal"
    seed: int = 1

    def scheduling_setup(self) -> ExperimentScheduling:
        n = parse_layout(self.layout).world_size
        return ExperimentScheduling(master_worker=TasksGroup(1, Scheduling(cpu=4, mem=16000)),
                                    model_worker=TasksGroup(n, Scheduling(cpu=8, gpu=1, mem=64000)))

    def initial_setup(self) -> ExperimentConfig:
        self.gen.n_samples = self.group_size
        actor_itf = ModelInterfaceAbstraction("ppo_actor", args=dict(
            generation_config=dataclasses.asdict(self.gen), group_size=self.group_size, n_minibatches=1))
        gen = MFCDef(name="actor_gen", n_seqs=self.eval_bs_n_seqs, interface_type=ModelInterfaceType.GENERATE,
                     interface_impl=actor_itf, model_name="actor", input_keys=("packed_prompts", "task_ids"),
                     output_keys=("packed_input_ids", "prompt_mask", "packed_logprobs", "seq_no_eos_mask",
                                  "version_start", "version_end"), mb_spec=self.mb_spec)
        rew = MFCDef(name="rew_inf", n_seqs=self.eval_bs_n_seqs, interface_type=ModelInterfaceType.INFERENCE,
                     interface_impl=ModelInterfaceAbstraction("rw-math-code", args=self.reward_args),
                     model_name="reward", input_keys=("packed_input_ids", "prompt_mask"), output_keys=("rewards",),
                     log_return_value=True, mb_spec=self.mb_spec)
        workers = symmetric_model_workers(
            {"actor": self.actor, "reward": ModelTrainEvalConfig(backend="null")}, parse_layout(self.layout),
            self.experiment_name, self.trial_name, self.device, self.fileroot,
            f"{self.fileroot}/param_realloc/{self.experiment_name}/{self.trial_name}", self.seed, [self.dataset],
            self.actor.tokenizer_path, shuffle_dataset=False)
        ctrl = ExperimentSaveEvalControl(total_train_epochs=1)
        return ExperimentConfig(exp_ctrl=ctrl, model_rpcs=[gen, rew], model_worker=workers, fileroot=self.fileroot)


register_experiment("math-code-eval", MathCodeEvalExperiment)
