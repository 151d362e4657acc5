"""Automatic evaluation of the checkpoints a trial saves.

``AutomaticEvaluator.step()`` scans the trial's checkpoint directory for ``epoch{e}epochstep{s}globalstep{g}``
folders, starts one evaluation sub-process per new checkpoint (at most `max_concurrent_jobs` at a time, in
global-step order), collects finished results and logs them (JSON lines + the stats logger sinks) exactly
once and in step order. Behaviour of ``realhf/scheduler/evaluator.py:25-345``.
"""
import dataclasses
import enum
import json
import os
import re
import subprocess
import sys
from typing import Callable, Dict, List, Optional

from ..utils import logging

logger = logging.getLogger("auto-evaluator")
CKPT_RE = re.compile(r"epoch(\d+)epochstep(\d+)globalstep(\d+)$")


class EvaluationStepStatus(enum.Enum):
    PENDING = 0
    RUNNING = 1
    FAILED = 2
    DONE = 3
    LOGGED = 4


@dataclasses.dataclass
class EvaluationStep:
    global_step: int
    ckpt_dir: str
    output_path: str
    status: EvaluationStepStatus = EvaluationStepStatus.PENDING
    process: Optional[subprocess.Popen] = None

    @staticmethod
    def from_ckpt_dir(ckpt_dir: str, eval_root: str) -> Optional["EvaluationStep"]:
        m = CKPT_RE.search(os.path.basename(ckpt_dir.rstrip("/")))
        if not m:
            return None
        g = int(m.group(3))
        return EvaluationStep(global_step=g, ckpt_dir=ckpt_dir, output_path=os.path.join(eval_root, f"globalstep{g}.json"))


@dataclasses.dataclass
class AutomaticEvaluatorConfig:
    data_path: str = ""
    n_sampling: int = 1
    max_new_tokens: int = 2048
    temperature: float = 0.6
    greedy: bool = False
    prompt_template: str = "{prompt}"
    max_concurrent_jobs: int = 1
    initial_checkpoint_path: Optional[str] = None
    device: Optional[str] = None


class AutomaticEvaluator:
    def __init__(self, config: AutomaticEvaluatorConfig, ckpt_root: str, eval_root: str,
                 log_fn: Optional[Callable[[Dict, int], None]] = None, cmd_builder: Optional[Callable] = None):
        self.config, self.ckpt_root, self.eval_root = config, ckpt_root, eval_root
        self.steps: Dict[int, EvaluationStep] = {}
        self.log_fn = log_fn
        self.cmd_builder = cmd_builder or self._default_cmd
        os.makedirs(eval_root, exist_ok=True)
        if config.initial_checkpoint_path:
            self.steps[0] = EvaluationStep(0, config.initial_checkpoint_path, os.path.join(eval_root, "globalstep0.json"))
        # results of an earlier run of the same trial are not recomputed
        for s in list(self._scan().values()):
            if os.path.exists(s.output_path):
                s.status = EvaluationStepStatus.LOGGED
                self.steps[s.global_step] = s

    def _default_cmd(self, step: EvaluationStep) -> List[str]:
        c = self.config
        cmd = [sys.executable, "-m", "areal_vllm_amd.evaluation.math_eval", "--model", step.ckpt_dir, "--data", c.data_path,
               "--n-sampling", str(c.n_sampling), "--max-new-tokens", str(c.max_new_tokens), "--temperature",
               str(c.temperature), "--prompt-template", c.prompt_template, "--output", step.output_path]
        if c.greedy:
            cmd.append("--greedy")
        if c.device:
            cmd += ["--device", c.device]
        return cmd

    def _scan(self) -> Dict[int, EvaluationStep]:
        found = {}
        if os.path.isdir(self.ckpt_root):
            for d in os.listdir(self.ckpt_root):
                s = EvaluationStep.from_ckpt_dir(os.path.join(self.ckpt_root, d), self.eval_root)
                if s is not None:
                    found[s.global_step] = s
        return found

    def step(self):
        for g, s in self._scan().items():
            self.steps.setdefault(g, s)
        running = [s for s in self.steps.values() if s.status == EvaluationStepStatus.RUNNING]
        for s in running:
            rc = s.process.poll()
            if rc is None:
                continue
            ok = rc == 0 and os.path.exists(s.output_path)
            s.status = EvaluationStepStatus.DONE if ok else EvaluationStepStatus.FAILED
            if not ok:
                logger.warning(f"evaluation of global step {s.global_step} failed (exit code {rc})")
        n_running = sum(s.status == EvaluationStepStatus.RUNNING for s in self.steps.values())
        for g in sorted(self.steps):
            s = self.steps[g]
            if n_running >= self.config.max_concurrent_jobs:
                break
            if s.status == EvaluationStepStatus.PENDING:
                s.process = subprocess.Popen(self.cmd_builder(s), stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
                s.status = EvaluationStepStatus.RUNNING
                n_running += 1
                logger.info(f"started evaluation of global step {g}")
        # log strictly in step order: a finished later step waits for the earlier ones
        for g in sorted(self.steps):
            s = self.steps[g]
            if s.status in (EvaluationStepStatus.LOGGED, EvaluationStepStatus.FAILED):
                continue
            if s.status != EvaluationStepStatus.DONE:
                break
            with open(s.output_path) as f:
                metrics = json.load(f).get("metrics", {})
            with open(os.path.join(self.eval_root, "eval_results.jsonl"), "a") as f:
                f.write(json.dumps(dict(global_step=g, **metrics)) + "\n")
            if self.log_fn is not None:
                self.log_fn({f"eval/{k}": v for k, v in metrics.items()}, g)
            s.status = EvaluationStepStatus.LOGGED
            logger.info(f"evaluation of global step {g}: {metrics}")

    def pending(self) -> bool:
        return any(s.status in (EvaluationStepStatus.PENDING, EvaluationStepStatus.RUNNING, EvaluationStepStatus.DONE)
                   for s in self.steps.values())
