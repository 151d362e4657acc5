"""Experiments with "null" models and the null interface: the whole worker runtime runs (master, model
workers, data-flow graph, data redistribution) but no model computes anything, which isolates runtime
overhead. Role of ``realhf/experiments/common/null_exp.py`` (``null-sft``, ``null-ppo``)."""
import dataclasses

from ..api.cli_args import MicroBatchSpec
from ..system.dfg import MFCDef, ModelInterfaceAbstraction, ModelInterfaceType
from ..system.system_api import (DatasetAbstraction, Experiment, ExperimentConfig, ExperimentSaveEvalControl,
                                 ExperimentScheduling, Scheduling, TasksGroup, register_experiment)
from .common import ModelTrainEvalConfig, parse_layout, symmetric_model_workers


@dataclasses.dataclass
class _NullBase(Experiment):
    experiment_name: str = "null"
    trial_name: str = "trial"
    layout: str = "d1t1p1"
    train_bs_n_seqs: int = 8
    mb_spec: MicroBatchSpec = dataclasses.field(default_factory=MicroBatchSpec)
    exp_ctrl: ExperimentSaveEvalControl = dataclasses.field(default_factory=ExperimentSaveEvalControl)
    tokenizer_path: str = ""
    device: str = "cpu"
    fileroot: str = "/tmp/areal"
    seed: int = 1

    def scheduling_setup(self) -> ExperimentScheduling:
        n = parse_layout(self.layout).world_size
        return ExperimentScheduling(master_worker=TasksGroup(1, Scheduling(cpu=4, mem=16000)),
                                    model_worker=TasksGroup(n, Scheduling(cpu=4, gpu=0, mem=16000)))

    def _workers(self, dataset):
        return symmetric_model_workers(
            {"default": ModelTrainEvalConfig(backend="null")}, parse_layout(self.layout), self.experiment_name,
            self.trial_name, self.device, self.fileroot,
            f"{self.fileroot}/param_realloc/{self.experiment_name}/{self.trial_name}", self.seed, [dataset],
            self.tokenizer_path or None)


@dataclasses.dataclass
class NullSFTExperiment(_NullBase):
    dataset: DatasetAbstraction = dataclasses.field(default_factory=lambda: DatasetAbstraction("prompt_answer"))

    def initial_setup(self) -> ExperimentConfig:
        rpc = MFCDef(name="trainDefault", n_seqs=self.train_bs_n_seqs, interface_type=ModelInterfaceType.TRAIN_STEP,
                     interface_impl=ModelInterfaceAbstraction("null"), model_name="default",
                     input_keys=("packed_input_ids", "prompt_mask"), log_return_value=True, mb_spec=self.mb_spec)
        return ExperimentConfig(exp_ctrl=self.exp_ctrl, model_rpcs=[rpc], model_worker=self._workers(self.dataset),
                                fileroot=self.fileroot)


@dataclasses.dataclass
class NullPPOExperiment(_NullBase):
    dataset: DatasetAbstraction = dataclasses.field(default_factory=lambda: DatasetAbstraction("math_code_prompt"))

    def initial_setup(self) -> ExperimentConfig:
        null = ModelInterfaceAbstraction("null")
        rw = MFCDef(name="reward", n_seqs=self.train_bs_n_seqs, interface_type=ModelInterfaceType.INFERENCE,
                    interface_impl=null, model_name="default", input_keys=("packed_prompts",),
                    output_keys=("rewards",), mb_spec=self.mb_spec)
        tr = MFCDef(name="trainDefault", n_seqs=self.train_bs_n_seqs, interface_type=ModelInterfaceType.TRAIN_STEP,
                    interface_impl=null, model_name="default", input_keys=("packed_prompts", "rewards"),
                    log_return_value=True, mb_spec=self.mb_spec)
        return ExperimentConfig(exp_ctrl=self.exp_ctrl, model_rpcs=[rw, tr], model_worker=self._workers(self.dataset),
                                fileroot=self.fileroot)


register_experiment("null-sft", NullSFTExperiment)
register_experiment("null-ppo", NullPPOExperiment)
