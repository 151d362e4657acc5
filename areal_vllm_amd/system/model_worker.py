"""Model worker: one process per accelerator hosting shards of one or more models.

All model workers of an experiment form one ``torch.distributed`` world (RCCL on MI355X, gloo on the
host); every model builds its own rank grid inside it, so an actor with d4 can share workers with a
reference model with d2t2. The worker serves the master's requests in arrival order:

  spec / model_config / initialize          start-up
  fetch                                     next dataset batch (or the trajectories pulled from rollout
                                            workers) into the local `DataManager`, metadata back
  inference / generate / train_step         a model function call on locally stored inputs; outputs stay
                                            here, the data-parallel head returns their metadata / statistics
  empty                                     no-op carrier of hooks
  clear_data_cache / save / evaluate / reset

and, around them, the hooks ``data_transfer`` (execute a redistribution plan), ``param_realloc`` (publish
new weights for the generation servers: HF checkpoint under the realloc root + a version bump in
name_resolve) and ``offload`` (park an idle model in pinned host memory). Behaviour of ``realhf/system/model_worker.py:101-1610``.
"""
import dataclasses
import os
import time
import traceback
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist

from ..api.io_struct import FinetuneSpec, SaveLoadMeta
from ..impl.dataset import make_dataset
from ..parallel.grid import DIST_TIMEOUT, ProcessGrid
from ..utils import logging, name_resolve, names, network, seeding, stats_tracker
from .data_api import DataBatchMeta, SequenceSample
from .data_manager import DataManager
from .dfg import MFCDef, ModelInterfaceType, ModelName
from .master_worker import MASTER_STREAM
from .model_api import Model, make_backend, make_interface, make_model
from .stream_dataset import PullerStreamDataset
from .streams import NameResolvingReplyServer, NoMessage, Payload
from .system_api import ModelShardID, ModelWorkerConfig, StandaloneModelShard
from .worker_base import PollResult, Worker

logger = logging.getLogger("model-worker")


class ModelWorker(Worker):
    def _configure(self, config: ModelWorkerConfig):
        self.config = config
        idx = self._info.worker_index
        if config.device == "cpu":
            os.environ["AREAL_PLATFORM"] = "cpu"
        seeding.set_random_seed(config.base_seed, f"model_worker{idx}")
        self.reply = NameResolvingReplyServer(self.experiment_name, self.trial_name, MASTER_STREAM, handler=idx)
        self.shards: List[StandaloneModelShard] = list(config.shards)
        self.models: Dict[ModelName, Model] = {}
        self.backends: Dict[ModelName, Any] = {}
        self.interfaces: Dict[str, Any] = {}  # rpc name -> interface
        self.rpcs: Dict[str, MFCDef] = {r.name: r for r in config.model_rpcs}
        self.grids: Dict[ModelName, ProcessGrid] = {}
        self.data_manager: Optional[DataManager] = None
        self.dataset = None
        self.dataloader = None
        self.stream_dataset: Optional[PullerStreamDataset] = None
        self._setup_done = False
        self._epoch = 0
        return self._info

    # --------------------------------------------------------------------------- start-up
    def _is_data_head(self) -> Optional[int]:
        """dp rank of the source model whose head this worker is (it loads data), else None."""
        cfg = self.config
        src = [r for r in cfg.model_rpcs if r.is_src]
        if not src:
            return None
        topo = cfg.model_topos[src[0].model_name]
        for d in range(topo.dp_size):
            if cfg.msid2mwid[ModelShardID(src[0].model_name, d, 0, topo.pp_size - 1)] == self.worker_index:
                return d
        return None

    def _init_world(self):
        cfg = self.config
        n, idx = cfg.n_model_workers, self.worker_index
        key = names.distributed_master(self.experiment_name, self.trial_name, "model_workers")
        if idx == 0:
            addr = f"{network.gethostip()}:{network.find_free_ports(1)[0]}"
            name_resolve.add(key, addr, replace=True)
        else:
            addr = name_resolve.wait(key, timeout=300)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if cfg.device == "cpu":
            self.device = torch.device("cpu")
            if addr.startswith("127."):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo", init_method=f"tcp://{addr}", rank=idx, world_size=n, timeout=DIST_TIMEOUT)
        else:
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            self.device = torch.device("cuda", local)
            dist.init_process_group("cpu:gloo,cuda:nccl", init_method=f"tcp://{addr}", rank=idx, world_size=n,
                                    timeout=DIST_TIMEOUT, device_id=self.device)

    def _lazy_setup(self):
        cfg = self.config
        if not dist.is_initialized():
            self._init_world()
        else:  # in-process tests may bring their own world
            self.device = torch.device("cpu" if cfg.device == "cpu" else f"cuda:{torch.cuda.current_device()}")
        # every worker builds every model's grid in the same order: group creation is collective
        for name in sorted(cfg.model_topos, key=str):
            topo = cfg.model_topos[name]
            ranks = [cfg.msid2mwid[ModelShardID(name, d, t, p)] for p in range(topo.pp_size)
                     for d in range(topo.dp_size) for t in range(topo.tp_size)]
            self.grids[name] = ProcessGrid(topo, ranks=ranks)
        self.data_manager = DataManager(self.worker_index, self.device)
        dp = self._is_data_head()
        if dp is not None and cfg.datasets:
            src = [r for r in cfg.model_rpcs if r.is_src][0]
            n_dp = cfg.model_topos[src.model_name].dp_size
            stream = [d for d in cfg.datasets if d.type_ == "puller_stream"]
            if stream:
                self.stream_dataset = PullerStreamDataset(self.experiment_name, self.trial_name, puller_index=dp,
                                                          **stream[0].args)
            else:
                sets = [make_dataset(d, cfg.base_seed, dp, n_dp, cfg.tokenizer_name_or_path) for d in cfg.datasets]
                self.dataset = sets[0] if len(sets) == 1 else torch.utils.data.ConcatDataset(sets)
                self._batch_size = max(1, src.n_seqs // n_dp)
                self._gen = torch.Generator().manual_seed(seeding.get_seed())
                self._new_epoch()
        self._setup_done = True

    def _new_epoch(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self._gen).tolist() if self.config.shuffle_dataset else list(range(n))
        self._order, self._cursor = order, 0

    def _initialize_models(self, ft_spec: FinetuneSpec, recover: bool, recover_step: Optional[Dict]):
        cfg = self.config
        for shard in sorted(self.shards, key=lambda s: str(s.id)):
            name = shard.id.model_name
            if not shard.should_instantiate:
                continue
            model = make_model(shard.model, name, self.device)
            model.grid = self.grids[name]
            backend = make_backend(shard.backend)
            model = backend.initialize(model, ft_spec)
            self.models[name], self.backends[name] = model, backend
            if recover:
                ckpt = os.path.join(cfg.fileroot, "checkpoints", self.experiment_name, self.trial_name, name.role,
                                    "recover_checkpoint")
                if os.path.isdir(ckpt) and shard.backend.type_ != "null" and \
                        getattr(model.module, "optimizer", None) is not None:
                    backend.load(model, ckpt)
                    logger.info(f"model {name} recovered from {ckpt}")
                if recover_step:
                    model.version.global_step = int(recover_step["global_step"])
                    model.version.epoch, model.version.epoch_step = int(recover_step["epoch"]), int(recover_step["epoch_step"])
            if shard.eval_dataset is not None:
                g = self.grids[name]
                model.eval_dataset = make_dataset(shard.eval_dataset, cfg.base_seed, g.coord.dp, g.dp,
                                                  cfg.tokenizer_name_or_path)
                model.eval_bs = shard.eval_bs
        for r in cfg.model_rpcs:
            if r.model_name in self.models:
                self.interfaces[r.name] = make_interface(r.interface_impl)
        # further replicas of a role start from replica 0's weights (every worker walks the same list: the
        # transfers are collective over the workers that host either side)
        for name in sorted(cfg.model_topos, key=str):
            first = ModelName(name.role, 0)
            if name.replica_id > 0 and first in cfg.model_topos:
                self._realloc_between_replicas(first, name, 1.0)

    # ----------------------------------------------------------------------------- handlers
    def _handle_spec(self, _):
        if self.stream_dataset is not None:
            return dict(dataset_size=len(self.stream_dataset), n_datasets=1)
        return dict(dataset_size=len(self.dataset) if self.dataset is not None else 0, n_datasets=1)

    def _handle_model_config(self, data):
        """Geometry for FLOP accounting (None for parameter-free roles)."""
        shard = next((s for s in self.shards if s.id.model_name == data["model_name"]), None)
        if shard is None or shard.model.type_ != "native":
            return None
        from ..engine.train_engine import _resolve_model_config

        mc = _resolve_model_config(shard.model.args["path"], shard.model.args.get("is_critic", False))
        return dict(num_layers=mc.n_layers, hidden_size=mc.hidden_size, intermediate_size=mc.intermediate_size,
                    vocab_size=mc.vocab_size, n_q_heads=mc.n_q_heads, n_kv_heads=mc.n_kv_heads, head_dim=mc.head_dim)

    def _handle_fetch(self, _) -> DataBatchMeta:
        dp = self._is_data_head()
        if self.stream_dataset is not None:
            items = self.stream_dataset.get()
            if not items:
                return DataBatchMeta(dp_rank=dp, meta_sample=None)
            x = SequenceSample.gather(items)
            self.data_manager.store(x)
            return DataBatchMeta(dp_rank=dp, meta_sample=x.meta())
        if self.dataset is None:
            return DataBatchMeta(dp_rank=dp if dp is not None else -1, meta_sample=None)
        final = False
        idx = self._order[self._cursor:self._cursor + self._batch_size]
        self._cursor += len(idx)
        if self._cursor >= len(self._order):
            final = True
            self._epoch += 1
            self._new_epoch()
        if not idx:
            return DataBatchMeta(dp_rank=dp, meta_sample=None, is_final_batch=final)
        x = SequenceSample.gather([self.dataset[i] for i in idx])
        self.data_manager.store(x)
        return DataBatchMeta(dp_rank=dp, meta_sample=x.meta(), is_final_batch=final)

    def _handle_mfc(self, handle_name: str, data: Dict) -> Any:
        rpc = self.rpcs[data["rpc_name"]]
        model = self.models[rpc.model_name]
        itf = self.interfaces[rpc.name]
        x = self.data_manager.get(data["ids"], rpc.input_keys) if rpc.input_keys else \
            SequenceSample(keys=set(), ids=data["ids"], seqlens={})
        if rpc.input_key_remap:
            x.remap_keys_(rpc.input_key_remap)
        tik = time.perf_counter()
        with self._maybe_profile_rpc(rpc):
            if handle_name == "train_step":
                res = itf.train_step(model, x, rpc.mb_spec)
            elif handle_name == "inference":
                res = itf.inference(model, x, rpc.mb_spec)
            elif handle_name == "generate":
                res = itf.generate(model, x, rpc.mb_spec)
            else:
                raise NotImplementedError(handle_name)
        self._log_gpu_stats(rpc.name)
        stats_tracker.export()  # interfaces report through their return value; drop stray local stats
        if isinstance(res, SequenceSample):
            if rpc.output_key_remap:
                res.remap_keys_(rpc.output_key_remap)
            res = res.select([k for k in res.keys if k in set(rpc.output_keys)] or list(res.keys))
            if not data["is_dp_head"]:
                return None
            self.data_manager.store(res)
            return res.meta()
        if not data["is_dp_head"]:
            return None
        if isinstance(res, dict):
            res = dict(res, mfc_time=time.perf_counter() - tik)
        return res

    # ------------------------------------------------------------------ tracing / monitoring
    def _maybe_profile_rpc(self, rpc: MFCDef):
        """``AREAL_DUMP_TRACE=1``: wrap the call in the PyTorch profiler and write a chrome trace per
        (rpc, worker) under ``{fileroot}/logs/.../trace``; ``AREAL_DUMP_MEMORY=1`` adds an allocator
        snapshot. Off by default (per-call profiler of ``realhf/system/model_worker.py:79-95,829-909``)."""
        import contextlib

        if os.environ.get("AREAL_DUMP_TRACE", "0") != "1":
            return contextlib.nullcontext()
        worker = self

        class _Ctx:
            def __enter__(self_):
                acts = [torch.profiler.ProfilerActivity.CPU]
                if worker.device.type == "cuda":
                    acts.append(torch.profiler.ProfilerActivity.CUDA)
                    if os.environ.get("AREAL_DUMP_MEMORY", "0") == "1":
                        torch.cuda.memory._record_memory_history()
                self_.prof = torch.profiler.profile(activities=acts, record_shapes=True, with_stack=False)
                self_.prof.__enter__()

            def __exit__(self_, *exc):
                self_.prof.__exit__(*exc)
                d = os.path.join(worker.config.fileroot, "logs", worker.experiment_name, worker.trial_name, "trace")
                os.makedirs(d, exist_ok=True)
                worker._trace_count = getattr(worker, "_trace_count", 0) + 1
                base = os.path.join(d, f"{rpc.name}_r{worker.worker_index}_{worker._trace_count}")
                self_.prof.export_chrome_trace(base + ".json")
                if worker.device.type == "cuda" and os.environ.get("AREAL_DUMP_MEMORY", "0") == "1":
                    torch.cuda.memory._dump_snapshot(base + "_mem.pkl")
                return False

        return _Ctx()

    def _log_gpu_stats(self, tag: str):
        """Memory high-water mark after a call; the worker takes itself down when the device is nearly
        full (``AREAL_GPU_MEMORY_KILL_THRESHOLD``, a fraction) so that the launcher's recover loop restarts
        the trial instead of the job dying later inside a collective."""
        if self.device.type != "cuda":
            return
        free, total = torch.cuda.mem_get_info(self.device)
        used = 1.0 - free / total
        peak = torch.cuda.max_memory_allocated(self.device) / 2 ** 30
        logger.debug(f"after {tag}: device memory {used * 100:.1f}% used, peak allocated {peak:.1f} GiB")
        thr = float(os.environ.get("AREAL_GPU_MEMORY_KILL_THRESHOLD", "1.0"))
        if used > thr:
            raise RuntimeError(f"device memory {used:.3f} above the kill threshold {thr}: exiting for a clean restart")

    def _hook_data_transfer(self, hd: Dict):
        self.data_manager.redistribute(hd["meta"], hd["plan"])

    def _hook_param_realloc(self, hd: Dict):
        """Hand the source model's weights on. With a `target` - another replica of the role hosted by model
        workers under its own (dp, tp, pp) layout - every destination shard pulls the slices it needs from the
        source shards that own them (``parallel/param_realloc.py``; the hook of
        ``realhf/system/model_worker.py:1046-1130``). Without one the weights are published for the
        generation fleet."""
        if hd.get("target") is not None:
            return self._realloc_between_replicas(hd["source"], hd["target"], float(hd.get("eta", 1.0)))
        name = hd["source"]
        model = self.models.get(name)
        if model is None:
            return
        v = model.version.global_step
        path = os.path.join(self.config.realloc_root, name.role, str(v))
        model.module.save(SaveLoadMeta(path=path, weight_format="hf", with_optim=False, tokenizer=model.tokenizer))
        if self.grids[name].local_index == 0:
            name_resolve.add(names.model_version(self.experiment_name, self.trial_name, name.role), str(v), replace=True)
            logger.info(f"published weights of {name} version {v} at {path}")

    def _realloc_between_replicas(self, source: ModelName, target: ModelName, eta: float):
        from ..parallel.param_realloc import Layout, derive_realloc_plan, realloc_parameters

        src_grid, dst_grid = self.grids[source], self.grids[target]
        src_model, dst_model = self.models.get(source), self.models.get(target)
        if src_model is None and dst_model is None:
            return
        eng = (src_model or dst_model).module
        for m_ in (src_model, dst_model):  # a parked model comes back before its buffer is read or written
            if m_ is not None and hasattr(m_.module, "onload"):
                m_.module.onload()
        mc = eng.model_config
        key = (source, target)
        plans = self.__dict__.setdefault("_realloc_plans", {})
        if key not in plans:
            assert src_grid.cp == dst_grid.cp == 1 and src_grid.ep == dst_grid.ep == 1, \
                "parameter re-allocation covers (dp, tp, pp) layouts"
            dtype = eng.model.flat.flat.dtype
            plans[key] = derive_realloc_plan(mc, Layout(src_grid.dp, src_grid.tp, src_grid.pp),
                                             Layout(dst_grid.dp, dst_grid.tp, dst_grid.pp), dtype=dtype)
        tik = time.perf_counter()
        n = realloc_parameters(plans[key], list(src_grid.ranks), list(dst_grid.ranks),
                               src_model.module.model.flat.flat if src_model is not None else None,
                               dst_model.module.model.flat.flat if dst_model is not None else None, eta=eta)
        if dst_model is not None:
            opt = getattr(dst_model.module, "optimizer", None)
            if opt is not None:  # a trainable target keeps its fp32 master copy in step with the new weights
                opt.resync_master_from_params()
            if src_model is not None:
                dst_model.version.global_step = src_model.version.global_step
            else:
                dst_model.version.global_step += 1
            logger.debug(f"{target} received {n} parameter elements from {source} in "
                         f"{(time.perf_counter() - tik) * 1e3:.1f} ms")
        if os.environ.get("AREAL_CHECK_REALLOC", "0") == "1" and eta == 1.0 and src_model is not None \
                and dst_model is not None:
            # debugging aid for layouts where every worker hosts a shard of both replicas: the two models, each
            # gathered over its own grid, must now be the same checkpoint
            from ..models import hf_io

            a = {k: v.clone() for k, v in hf_io.native_to_hf_state(src_model.module.model)}
            b = {k: v.clone() for k, v in hf_io.native_to_hf_state(dst_model.module.model)}
            common = sorted(set(a) & set(b))  # pipeline stages of the two layouts own different layers
            assert common, "the two replicas share no tensor on this worker"
            for k in common:
                assert torch.equal(a[k], b[k]), f"{target} differs from {source} in {k} after the re-allocation"
            self._realloc_checked = getattr(self, "_realloc_checked", 0) + 1
            logger.info(f"re-allocation {source} -> {target} verified on {len(common)} tensors")

    def _handle_save(self, data: Dict):
        model = self.models.get(data["model_name"])
        if model is None:
            return None
        if data.get("with_optim"):
            self.backends[data["model_name"]].save(model, data["save_dir"])
        else:
            itf = next((self.interfaces[r.name] for r in self.rpcs.values()
                        if r.model_name == data["model_name"] and r.interface_type == ModelInterfaceType.TRAIN_STEP), None)
            if itf is not None:
                itf.save(model, data["save_dir"])
        return data["save_dir"]

    def _handle_evaluate(self, data: Dict):
        model = self.models.get(data["model_name"])
        ds = getattr(model, "eval_dataset", None) if model is not None else None
        if ds is None:
            return None
        bs = getattr(model, "eval_bs", 128)
        loader = (SequenceSample.gather([ds[i] for i in range(s, min(len(ds), s + bs))]) for s in range(0, len(ds), bs))
        itf = next(self.interfaces[r.name] for r in self.rpcs.values() if r.model_name == data["model_name"])
        return itf.evaluate(model, loader)

    def _hook_offload(self, hd: Dict[str, Any]):
        """Post-hook of a model function call on a model configured with `offload`: park its weights (and
        optimizer state) in pinned host memory until the next call that needs them; the engine brings them back
        by itself. Models without that ability (null / mock backends) ignore the hook."""
        model = self.models.get(hd["model_name"])
        eng = getattr(model, "module", None)
        if eng is not None and hasattr(eng, "offload"):
            freed = eng.offload()
            if freed:
                logger.info(f"{hd['model_name']}: {freed / 2**30:.2f} GiB parked in host memory")

    def _dispatch(self, p: Payload) -> Any:
        h = p.handle_name
        if h == "empty":
            return None
        if h == "spec":
            return self._handle_spec(p.data)
        if h == "model_config":
            return self._handle_model_config(p.data)
        if h == "initialize":
            self._initialize_models(p.data["ft_spec"], p.data.get("recover", False), p.data.get("recover_step"))
            return None
        if h == "fetch":
            return self._handle_fetch(p.data)
        if h in ("train_step", "inference", "generate"):
            return self._handle_mfc(h, p.data)
        if h == "clear_data_cache":
            self.data_manager.clear_data(p.data)
            return None
        if h == "save":
            return self._handle_save(p.data)
        if h == "evaluate":
            return self._handle_evaluate(p.data)
        if h == "reset":
            self.exit()
            return None
        raise NotImplementedError(f"unknown request {h}")

    # -------------------------------------------------------------------------------- poll
    def _poll(self) -> PollResult:
        if not self._setup_done:
            self._lazy_setup()
        try:
            p = self.reply.poll(block=True, timeout=0.05)
        except NoMessage:
            return PollResult(0, 0)
        try:
            for hook, hd in zip(p.pre_hooks, p.pre_hook_data):
                getattr(self, f"_hook_{hook}")(hd)
            res = self._dispatch(p)
            for hook, hd in zip(p.post_hooks, p.post_hook_data):
                getattr(self, f"_hook_{hook}")(hd)
        except Exception as e:
            logger.error(f"request {p.handle_name} failed: {traceback.format_exc()}")
            self.reply.post(Payload(handler="master", handle_name=p.handle_name, request_id=p.request_id,
                                    data=dict(__error__=repr(e))))
            raise
        self.reply.post(Payload(handler="master", handle_name=p.handle_name, request_id=p.request_id, data=res))
        n = len(p.data["ids"]) if isinstance(p.data, dict) and "ids" in p.data else 0
        return PollResult(sample_count=n, batch_count=int(n > 0))

    def _exit_hook(self, exit_status):
        if self.stream_dataset is not None:
            self.stream_dataset.close()
        self.reply.close()
        if dist.is_initialized() and getattr(self, "_own_world", True):
            try:
                dist.destroy_process_group()
            except Exception:
                pass
