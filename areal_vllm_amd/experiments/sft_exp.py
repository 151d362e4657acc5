"""Supervised fine-tuning experiment on the worker runtime: one `train_step` MFC over a prompt-answer
dataset. Role of ``realhf/experiments/common/sft_exp.py``."""
import dataclasses
from typing import Any, Dict, Optional

from ..api.cli_args import MicroBatchSpec
from ..system.dfg import MFCDef, ModelInterfaceAbstraction, ModelInterfaceType
from ..system.system_api import (DatasetAbstraction, Experiment, ExperimentConfig, ExperimentSaveEvalControl,
                                 ExperimentScheduling, Scheduling, TasksGroup, register_experiment)
from .common import resolve_allocation_mode, ModelTrainEvalConfig, parse_layout, symmetric_model_workers


@dataclasses.dataclass
class SFTExperiment(Experiment):
    experiment_name: str = "sft"
    trial_name: str = "trial"
    model: ModelTrainEvalConfig = dataclasses.field(default_factory=ModelTrainEvalConfig)
    allocation_mode: Optional[str] = None  # e.g. "sglang.d4p1m1+d2p2m1"; overrides layout / server counts
    layout: str = "d1t1p1"
    dataset: DatasetAbstraction = dataclasses.field(default_factory=lambda: DatasetAbstraction("prompt_answer"))
    eval_dataset: Optional[DatasetAbstraction] = None
    train_bs_n_seqs: int = 8
    mb_spec: MicroBatchSpec = dataclasses.field(default_factory=MicroBatchSpec)
    exp_ctrl: ExperimentSaveEvalControl = dataclasses.field(default_factory=ExperimentSaveEvalControl)
    device: str = "cuda"
    fileroot: str = "/tmp/areal"
    seed: int = 1

    def scheduling_setup(self) -> ExperimentScheduling:
        resolve_allocation_mode(self)
        n = parse_layout(self.layout).world_size
        return ExperimentScheduling(master_worker=TasksGroup(1, Scheduling(cpu=4, mem=16000)),
                                    model_worker=TasksGroup(n, Scheduling(cpu=8, gpu=1, mem=64000)))

    def initial_setup(self) -> ExperimentConfig:
        resolve_allocation_mode(self)
        rpc = MFCDef(name="trainDefault", n_seqs=self.train_bs_n_seqs, interface_type=ModelInterfaceType.TRAIN_STEP,
                     interface_impl=ModelInterfaceAbstraction("sft"), model_name="default",
                     input_keys=("packed_input_ids", "prompt_mask"), log_return_value=True, mb_spec=self.mb_spec)
        workers = symmetric_model_workers(
            {"default": self.model}, parse_layout(self.layout), self.experiment_name, self.trial_name, self.device,
            self.fileroot, f"{self.fileroot}/param_realloc/{self.experiment_name}/{self.trial_name}", self.seed,
            [self.dataset], self.model.tokenizer_path, eval_dataset=self.eval_dataset, eval_role="default")
        return ExperimentConfig(exp_ctrl=self.exp_ctrl, model_rpcs=[rpc], model_worker=workers, fileroot=self.fileroot)


register_experiment("sft", SFTExperiment)
