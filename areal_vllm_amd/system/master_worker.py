"""Master worker: owns the data-flow graph, the sequence buffer and the experiment's clock.

Every poll is one training step: decide whether this step saves / checkpoints / evaluates, traverse the
MFC graph once through the `FunctionExecutor`, log timing and throughput, bump the step counters, dump
recover information after checkpoints and finish the experiment (status COMPLETE in name_resolve, which
rollout workers and the generation-server manager watch) once the configured epochs or benchmark steps
are done. Behaviour of ``realhf/system/master_worker.py:49-606``.
"""
import asyncio
import copy
import dataclasses
import json
import os
import time
from typing import Dict, List, Optional

import numpy as np

from ..api.io_struct import FinetuneSpec, StepInfo
from ..base import monitor
from ..utils import logging, name_resolve, names, timeutil
from .buffer import AsyncIOSequenceBuffer
from .dfg import MFCDef, ModelInterfaceType, build_graph
from .function_executor import FunctionExecutor
from .model_function_call import RPCCoroutineControl, model_workers_of
from .recover import RecoverInfo, dump_recover_info, load_recover_info
from .streams import NameResolvingRequestClient
from .system_api import MasterWorkerConfig
from .worker_base import AsyncWorker, ExpStatus, PollResult

logger = logging.getLogger("master-worker")
MASTER_STREAM = "master"


class MasterWorker(AsyncWorker):
    def _configure(self, config: MasterWorkerConfig):
        self.config = config
        self.rpcs: List[MFCDef] = config.model_rpcs
        if any(r._G is None for r in self.rpcs):
            build_graph(self.rpcs)
        self.src_rpc = [r for r in self.rpcs if r.is_src][0]
        ec = config.exp_ctrl
        self.save_ctl = timeutil.EpochStepTimeFreqCtl(ec.save_freq_epochs, ec.save_freq_steps, ec.save_freq_secs)
        self.ckpt_ctl = timeutil.EpochStepTimeFreqCtl(ec.ckpt_freq_epochs, ec.ckpt_freq_steps, ec.ckpt_freq_secs)
        self.eval_ctl = timeutil.EpochStepTimeFreqCtl(ec.eval_freq_epochs, ec.eval_freq_steps, ec.eval_freq_secs)
        self.step_info = StepInfo(epoch=0, epoch_step=0, global_step=0, steps_per_epoch=1)
        self.last_step_info = copy.deepcopy(self.step_info)
        self.ctrl = RPCCoroutineControl()
        self.e2e_time_history: List[float] = []
        self._initialized = False
        self.stream: Optional[NameResolvingRequestClient] = None
        self.func_executor: Optional[FunctionExecutor] = None
        self.recover_run, self.recover_info = load_recover_info(self._recover_dir())
        if self.recover_run:
            ri = self.recover_info
            self.step_info = copy.deepcopy(ri.recover_start)
            self.last_step_info = copy.deepcopy(ri.last_step_info)
            self.save_ctl.load_state_dict(ri.save_ctl_info)
            self.ckpt_ctl.load_state_dict(ri.ckpt_ctl_info)
            self.eval_ctl.load_state_dict(ri.eval_ctl_info)
            logger.info(f"Recovering from step {self.step_info}; {len(ri.hash_vals_to_ignore)} samples of the "
                        f"interrupted epoch were already used")
        self._train_start = time.perf_counter()
        return self._info

    def _recover_dir(self) -> str:
        return os.path.join(self.config.fileroot, "recover", self.experiment_name, self.trial_name)

    def _save_root(self) -> str:
        return os.path.join(self.config.fileroot, "checkpoints", self.experiment_name, self.trial_name)

    # --------------------------------------------------------------------------- start-up
    def _lazy_init(self):
        cfg = self.config
        handlers = list(range(cfg.n_model_workers))
        self.stream = NameResolvingRequestClient(self.experiment_name, self.trial_name, MASTER_STREAM, handlers)
        topo = cfg.model_topos[self.src_rpc.model_name]
        _, _, data_heads = model_workers_of(self.src_rpc.model_name, topo, cfg.msid2mwid)
        specs = self.stream.call(data_heads, "spec", [None] * len(data_heads))
        self.dataset_size = int(sum(s["dataset_size"] for s in specs))
        train_rpcs = [r for r in self.rpcs if r.interface_type == ModelInterfaceType.TRAIN_STEP]
        bs = train_rpcs[0].n_seqs if train_rpcs else self.src_rpc.n_seqs
        assert all(r.n_seqs == bs for r in train_rpcs), "all train steps must use the same batch size"
        self.ft_spec = FinetuneSpec(total_train_epochs=cfg.exp_ctrl.total_train_epochs,
                                    dataset_size=max(self.dataset_size, bs), train_batch_size=bs)
        self.steps_per_epoch = max(1, self.dataset_size // self.src_rpc.n_seqs)
        self.step_info.steps_per_epoch = self.steps_per_epoch
        self.model_configs = {}
        for name, t in cfg.model_topos.items():
            ws, _, heads = model_workers_of(name, t, cfg.msid2mwid)
            self.model_configs[name] = self.stream.call([heads[0]], "model_config", [dict(model_name=name)])[0]
        # every model worker initialises all of its shards (process groups are collective)
        self.stream.call(handlers, "initialize", [dict(ft_spec=self.ft_spec, recover=self.recover_run,
                                                       recover_step=dataclasses.asdict(self.step_info))] * len(handlers))
        self.buffer = AsyncIOSequenceBuffer(self.rpcs, max_size=int(os.getenv("AREAL_MASTER_BUFFER_SIZE",
                                                                              str(cfg.buffer_size))))
        self.func_executor = FunctionExecutor(self.rpcs, cfg.msid2mwid, self.stream, self.buffer, cfg.model_topos,
                                              self.ctrl, cfg.n_model_workers, self.experiment_name, self.trial_name)
        if self.recover_run:
            self.func_executor.ignore_ids(self.recover_info.hash_vals_to_ignore)
            self.ctrl.used_ids_this_epoch = set(self.recover_info.hash_vals_to_ignore)
        self._initialized = True
        self._train_start = time.perf_counter()
        logger.info(f"master worker ready: dataset size {self.dataset_size}, {self.steps_per_epoch} steps/epoch, "
                    f"{len(self.rpcs)} model function calls")

    # -------------------------------------------------------------------------------- poll
    async def _poll_async(self) -> PollResult:
        key = names.experiment_status(self.experiment_name, self.trial_name)
        name_resolve.add(key, ExpStatus.RUNNING.value, replace=True)
        try:
            return await self._step()
        except Exception:
            name_resolve.add(key, ExpStatus.ABORTED.value, replace=True)
            raise

    async def _step(self) -> PollResult:
        if not self._initialized:
            self._lazy_init()
        si = self.step_info
        tik = time.perf_counter()
        is_epoch_last_step = si.epoch_step == self.steps_per_epoch - 1
        should_eval = self.eval_ctl.check(epochs=int(is_epoch_last_step), steps=1)
        should_save = self.save_ctl.check(epochs=int(is_epoch_last_step), steps=1)
        should_ckpt = self.ckpt_ctl.check(epochs=int(is_epoch_last_step), steps=1)
        logger.info(f"The next step is epoch {si.epoch + 1}/{self.config.exp_ctrl.total_train_epochs} step "
                    f"{si.epoch_step + 1}/{self.steps_per_epoch} (global step {si.global_step + 1}). Should "
                    f"checkpoint? {should_ckpt}. Should save? {should_save}. Should run evaluation? {should_eval}.")
        stats = await self.func_executor.execute_step_async()
        if should_save or should_ckpt:
            await self._save_models(recover=should_ckpt and not should_save)
            self.last_step_info = copy.deepcopy(si)
        if should_eval:
            await self._evaluate_models()
        if is_epoch_last_step:
            self.ctrl.used_ids_this_epoch.clear()
        e2e = time.perf_counter() - tik
        self.e2e_time_history.append(e2e)
        self._log_training_stats(e2e, stats)
        self.step_info = si.next()
        if should_save or should_ckpt:
            self._recover_save()
        ec = self.config.exp_ctrl
        total_steps = ec.total_train_epochs * self.steps_per_epoch
        if (ec.benchmark_steps is not None and self.step_info.global_step >= ec.benchmark_steps) or \
                self.step_info.global_step >= total_steps or \
                (ec.benchmark_n_seqs is not None and
                 self.step_info.global_step * self.ft_spec.train_batch_size >= ec.benchmark_n_seqs):
            return self.experiment_complete_exit()
        return PollResult(sample_count=self.src_rpc.n_seqs, batch_count=1)

    # -------------------------------------------------------------------- save / eval / log
    def _train_models(self):
        cfg = self.config
        out = []
        for r in self.rpcs:
            if r.interface_type == ModelInterfaceType.TRAIN_STEP and r.model_name not in [m for m, _ in out]:
                ws, _, _ = model_workers_of(r.model_name, cfg.model_topos[r.model_name], cfg.msid2mwid)
                out.append((r.model_name, ws))
        return out

    async def _save_models(self, recover: bool):
        si = self.step_info
        loop = asyncio.get_running_loop()
        for name, ws in self._train_models():
            tag = "recover_checkpoint" if recover else \
                f"epoch{si.epoch + 1}epochstep{si.epoch_step + 1}globalstep{si.global_step + 1}"
            path = os.path.join(self._save_root(), name.role, tag)
            await loop.run_in_executor(None, lambda: self.stream.call(
                ws, "save", [dict(model_name=name, save_dir=path, with_optim=recover)] * len(ws)))
            logger.info(f"saved model {name} to {path}")

    async def _evaluate_models(self):
        loop = asyncio.get_running_loop()
        for name, ws in self._train_models():
            res = await loop.run_in_executor(None, lambda: self.stream.call(
                ws, "evaluate", [dict(model_name=name)] * len(ws)))
            res = [r for r in res if r]
            if res:
                logger.info(f"evaluation of {name}: {res[0]}")

    def _log_training_stats(self, e2e: float, stats: Dict):
        si = self.step_info
        s = (f"Epoch {si.epoch + 1}/{self.config.exp_ctrl.total_train_epochs} step {si.epoch_step + 1}/"
             f"{self.steps_per_epoch} (global step {si.global_step + 1}) finishes. #End to end# execution time: "
             f"*{e2e:.3f}*s. Total time consumption: {time.perf_counter() - self._train_start:.3f}s. ")
        if len(self.e2e_time_history) > 2:
            avg = float(np.mean(self.e2e_time_history[2:]))
            total = self.config.exp_ctrl.total_train_epochs * self.steps_per_epoch
            s += f"Estimated remaining time: {avg * max(0, total - si.global_step - 1):.3f}s. "
        flops = 0.0
        for name, st in stats.items():
            res = st.get("result")
            n_tok = None
            if isinstance(res, dict):
                n_tok = res.get("n_tokens")
            elif isinstance(res, list) and res and isinstance(res[0], dict):
                n_tok = sum(x.get("n_tokens", 0) for x in res)
            rpc = next(r for r in self.rpcs if r.name == name)
            mc = self.model_configs.get(rpc.model_name)
            if n_tok and mc:
                f = monitor.transformer_forward_flops(n_tokens=n_tok, avg_seqlen=max(1.0, n_tok / max(1, st["n_seqs"])),
                                                      **mc)
                flops += f * (3 if rpc.interface_type == ModelInterfaceType.TRAIN_STEP else 1)
        if flops > 0:
            s += (f"TFLOP/s per GPU: {flops / e2e / self.config.n_model_workers / 1e12:.2f}, total TFLOP/s: "
                  f"{flops / e2e / 1e12:.2f}. ")
        logger.info(s + " | ".join(f"{k}: {v['time']:.2f}s" for k, v in stats.items()))
        log_dir = os.path.join(self.config.fileroot, "logs", self.experiment_name, self.trial_name)
        os.makedirs(log_dir, exist_ok=True)
        with open(os.path.join(log_dir, "master_stats.jsonl"), "a") as f:
            row = dict(global_step=si.global_step + 1, e2e=e2e)
            for k, v in stats.items():
                row[f"{k}/time"] = v["time"]
                res = v.get("result")
                if isinstance(res, list) and res and isinstance(res[-1], dict):
                    merged = {}
                    for part in reversed(res):  # batch-level statistics live in the first minibatch's dict
                        merged.update(part)
                    res = merged
                if isinstance(res, dict):
                    row.update({f"{k}/{kk}": vv for kk, vv in res.items() if isinstance(vv, (int, float))})
            f.write(json.dumps(row) + "\n")

    def _recover_save(self):
        info = RecoverInfo(recover_start=copy.deepcopy(self.step_info), last_step_info=self.last_step_info,
                           save_ctl_info=self.save_ctl.state_dict(), ckpt_ctl_info=self.ckpt_ctl.state_dict(),
                           eval_ctl_info=self.eval_ctl.state_dict(),
                           hash_vals_to_ignore=sorted(self.ctrl.used_ids_this_epoch))
        dump_recover_info(self._recover_dir(), info)
        logger.info(f"Dumped recover info; will recover from {info.recover_start} "
                    f"({len(info.hash_vals_to_ignore)} samples used in this epoch)")

    def experiment_complete_exit(self) -> PollResult:
        logger.info("Experiment completes.")
        name_resolve.add(names.experiment_status(self.experiment_name, self.trial_name), ExpStatus.COMPLETE.value,
                         replace=True)
        ws = list(range(self.config.n_model_workers))
        try:
            self.stream.call(ws, "reset", [None] * len(ws))
        except Exception:  # workers may already be gone
            pass
        self.stream.close()
        self._initialized = False
        self.exit()
        return PollResult(0, 0)
