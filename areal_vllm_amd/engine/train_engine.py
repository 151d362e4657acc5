"""Native SPMD train engine: packed micro-batches -> HIP-kernel transformer -> ZeRO-sharded AdamW.

Implements the reference's ``TrainEngine`` contract (``areal/api/engine_api.py:39-155``) with the
behaviour of its FSDP engine (``areal/engine/base_hf_engine.py:116-558``, ``areal/engine/fsdp_engine.py:
265-729``): token-budget micro-batching, global-token-mean loss scaling, grad-norm clipping with
skip-on-non-finite, HF-format save/load, per-rank optimizer state files, weight upload to rollout
engines (disk / collective / in-process). Parallelism (dp x cp x tp, pipeline stages via
``parallel/pipeline.py``) comes from the allocation mode.

Loss functions receive a ``LogitsProxy`` instead of a dense ``[T, V]`` tensor: ``proxy.logprobs_entropy(
labels, temperature)`` runs the fused LM-head + log-prob kernel, ``proxy.materialize()`` returns dense
logits for loss functions written against the reference's signature.
"""
import dataclasses
import gc
import math
import os
import time
from concurrent.futures import Future
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .. import ops
from ..api.alloc_mode import ParallelStrategy
from ..api.cli_args import MicroBatchSpec, OptimizerConfig, TrainEngineConfig
from ..api.engine_api import Scheduling, TrainEngine
from ..api.io_struct import FinetuneSpec, ParamSpec, SaveLoadMeta, WeightUpdateMeta
from ..models import hf_io
from ..models.config import PRESETS, ModelConfig, preset
from ..models.transformer import ALIGN, FlatParams, Transformer
from ..ops import peer_collectives
from ..parallel import comm
from ..parallel.grid import ProcessGrid, init_distributed
from ..parallel.zero import ShardedAdamW, lr_lambda
from ..utils import datapack, logging, name_resolve, names
from ..utils.data import (
    MicroBatchList,
    amend_position_ids,
    pad_mb_list,
    repad_packed,
    split_padded_tensor_dict_into_mb_list,
    unpack_sequence,
)

logger = logging.getLogger("train-engine")


class LogitsProxy:
    """Lazy view of the LM-head output for one packed micro-batch.

    `hidden` holds the final-norm states of the PADDED packed stream (this rank's token slice when
    `sp`, a Ulysses context, is given). Labels and results are in un-padded packed coordinates: with
    sequence parallelism the labels are padded, sliced to the local tokens, and the per-token results
    are all-gathered back, so loss functions never see the partition."""

    def __init__(self, model: Transformer, hidden: torch.Tensor, chunk: int, keep_logits: bool, tp_group=None,
                 sp=None, pad_len: int = 0):
        self.model = model
        self.sp = sp
        self.pad_len = pad_len
        if sp is None and pad_len > 0:
            hidden = hidden[: hidden.shape[0] - pad_len]
        self.hidden = hidden
        self.chunk = chunk
        self.keep_logits = keep_logits
        self.tp_group = tp_group
        self._dense = None
        self.n_tokens = hidden.shape[0] * (sp.size if sp is not None else 1) - (pad_len if sp is not None else 0)

    @property
    def shape(self):
        return (self.n_tokens, 1 if self.model.cfg.is_critic else self.model.cfg.vocab_size)

    @property
    def device(self):
        return self.hidden.device

    def _localize(self, t: torch.Tensor) -> torch.Tensor:
        if self.sp is None:
            return t
        if self.pad_len:
            t = torch.nn.functional.pad(t, (0, self.pad_len))
        return self.sp.local_slice(t)

    def _globalize(self, t: torch.Tensor) -> torch.Tensor:
        if self.sp is None:
            return t
        return self.sp.gather_tokens(t)[: self.n_tokens]

    def logprobs_entropy(self, labels: torch.Tensor, temperature: float = 1.0):
        lp, ent = self.model.logprobs(self.hidden, self._localize(labels), temperature, self.chunk, self.keep_logits)
        return self._globalize(lp), self._globalize(ent)

    def logprobs(self, labels: torch.Tensor, temperature: float = 1.0):
        return self.logprobs_entropy(labels, temperature)[0]

    def values(self):
        return self._globalize(self.model.values(self.hidden))

    def materialize(self) -> torch.Tensor:
        if self._dense is None:
            self._dense = self._globalize(self.model.logits(self.hidden))
        return self._dense


def _resolve_model_config(path: str, is_critic: bool) -> ModelConfig:
    if path and os.path.isdir(path) and os.path.exists(os.path.join(path, "config.json")):
        return ModelConfig.from_pretrained(path, is_critic=is_critic)
    key = (path or "").lower()
    if key in PRESETS:
        return preset(key, is_critic=is_critic)
    aliases = {"qwen/qwen2.5-1.5b-instruct": "qwen2-1.5b", "qwen/qwen2-1.5b-instruct": "qwen2-1.5b",
               "qwen/qwen2-1.5b": "qwen2-1.5b", "qwen/qwen2.5-1.5b": "qwen2-1.5b",
               "qwen/qwen2.5-7b-instruct": "qwen2.5-7b", "qwen/qwen2.5-7b": "qwen2.5-7b",
               "qwen/qwen2.5-32b-instruct": "qwen2.5-32b", "qwen/qwen2.5-32b": "qwen2.5-32b"}
    if key in aliases:
        return preset(aliases[key], is_critic=is_critic)
    raise FileNotFoundError(f"model path '{path}' is neither a HF checkpoint directory nor a known preset")


class NativeTrainEngine(TrainEngine):
    def __init__(self, config: TrainEngineConfig):
        self.config = config
        self.optimizer_config: Optional[OptimizerConfig] = config.optimizer
        self.model: Optional[Transformer] = None
        self.optimizer: Optional[ShardedAdamW] = None
        self.expert_optimizer: Optional[ShardedAdamW] = None  # routed experts under expert parallelism
        self.grid: Optional[ProcessGrid] = None
        self.model_config: Optional[ModelConfig] = None
        self.is_critic = False
        self._version = 0
        self._lr_step = 0
        self._total_steps = 1
        self.initialized = False
        self.own_global_group = False
        self.device = torch.device("cpu")
        self._weight_update_group = None
        self._local_rollout_engine = None
        self._training = True
        self.zero3 = None  # parallel/zero3.Zero3 when cfg.native.zero_stage == 3

    # ------------------------------------------------------------------ process groups
    def create_process_group(self, parallel_strategy: Optional[ParallelStrategy] = None,
                             grid: Optional[ProcessGrid] = None):
        """`grid`: a ready-made rank grid over a SUBSET of the world (worker runtime: several models share
        one world of model workers, each with its own layout); otherwise the strategy must fill the world."""
        from ..platforms import current_platform

        if not dist.is_initialized():
            init_distributed()
            self.own_global_group = True
        if current_platform.device_type == "cuda":
            self.device = torch.device("cuda", torch.cuda.current_device())
        if grid is not None:
            assert grid.member, "this process is not part of the model's rank grid"
            self.parallel_strategy = grid.strategy
            self.grid = grid
            self.rank = grid.local_index
            self.world_size = grid.world
            return
        if parallel_strategy is None:
            parallel_strategy = ParallelStrategy(data_parallel_size=dist.get_world_size())
        vpp = getattr(parallel_strategy, "virtual_pipeline_parallel_size", None) or 1
        if vpp > 1 and parallel_strategy.pp_size == 1:
            raise ValueError("virtual pipeline stages need pipeline parallelism (pipeline_parallel_size > 1)")
        self.parallel_strategy = parallel_strategy
        assert parallel_strategy.world_size == dist.get_world_size(), (
            f"parallel strategy {parallel_strategy} needs {parallel_strategy.world_size} ranks, "
            f"world has {dist.get_world_size()}")
        self.grid = ProcessGrid(parallel_strategy)
        self.rank = dist.get_rank()
        self.world_size = dist.get_world_size()

    @property
    def data_parallel_group(self):
        return self.grid.group("dp")

    @property
    def data_parallel_rank(self) -> int:
        return self.grid.rank_in("dp")

    @property
    def data_parallel_world_size(self) -> int:
        return self.grid.dp

    def current_data_parallel_head(self) -> int:
        return self.grid.dp_head_global_rank()

    def is_data_parallel_head(self) -> bool:
        return self.grid.is_dp_head()

    @property
    def context_and_model_parallel_group(self):
        return self.grid.group("mp")

    def get_scheduling_config(self) -> Scheduling:
        return Scheduling(cpu=4, gpu=1, mem=32 * 1024, type="worker")

    # --------------------------------------------------------------------- initialisation
    def initialize(self, addr: Optional[str] = None, ft_spec: Optional[FinetuneSpec] = None):
        if self.grid is None:
            self.create_process_group()
        cfg = self.config
        dtype = getattr(torch, cfg.dtype)
        mc = _resolve_model_config(cfg.path, self.is_critic)
        if cfg.disable_dropout:  # RL recipes train without dropout whatever the checkpoint's config says
            mc = dataclasses.replace(mc, embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0)
        self.model_config = mc
        g = self.grid
        from ..utils import rng

        # dropout masks: different per data-parallel replica, shared inside a tensor-parallel group except where the
        # activation is sharded over it (utils/rng.py)
        rng.keys.configure(seed=int(getattr(cfg, "dropout_seed", 0) or 0) + 7919, replica=g.coord.dp * g.cp + g.coord.cp,
                           tp_rank=g.coord.tp)
        tik = time.perf_counter()
        lr, chunks = (0, mc.n_layers), None
        if g.pp > 1:
            from ..parallel.pipeline import stage_layer_range

            if g.vpp > 1:
                # interleaved stages: this rank hosts virtual stages pp_rank, pp_rank + pp, ... of pp * vpp
                assert mc.n_layers >= g.pp * g.vpp, "fewer layers than virtual pipeline stages"
                lr = None
                chunks = [stage_layer_range(mc.n_layers, g.pp * g.vpp, c * g.pp + g.coord.pp) for c in range(g.vpp)]
            else:
                lr = stage_layer_range(mc.n_layers, g.pp, g.coord.pp)
        self.model = Transformer(
            mc, dtype=dtype, device=self.device, tp_size=g.tp, tp_rank=g.coord.tp, tp_group=g.group("tp"),
            layer_range=lr, layer_chunks=chunks, has_embed=g.coord.pp == 0, has_head=g.coord.pp == g.pp - 1,
            ep_size=g.ep, ep_rank=g.rank_in("ep"), ep_group=g.group("ep"), etp_size=g.etp,
            gradient_checkpointing=cfg.gradient_checkpointing,
            sequence_parallel=cfg.native.sequence_parallel and g.tp > 1,
            vocab_parallel=g.tp > 1,
        )
        assert not (self.model.seq_par and g.cp > 1), "choose tensor-parallel sequence parallelism or cp, not both"
        if g.cp > 1:
            if cfg.native.context_parallel_mode == "zigzag":
                from ..parallel.context_parallel import ZigZagContext

                self.model.sp = ZigZagContext(g.group("cp"), g.cp, g.coord.cp)
            else:
                from ..parallel.ulysses import UlyssesContext

                self.model.sp = UlyssesContext(g.group("cp"), g.cp, g.coord.cp)
        use_peer = self.device.type == "cuda" and (cfg.native.peer_collectives or peer_collectives.ENABLED)
        if use_peer:
            # activation collectives of the compute stream: Ulysses exchange (cp), sequence parallelism (tp)
            for name in (("cp",) if g.cp > 1 else ()) + (("tp",) if self.model.seq_par else ()):
                if comm.peer(g.group(name)) is None:
                    comm.attach_peer(g.group(name), peer_collectives.open_for(g.group(name), self.device))
            if self.model.seq_par and os.environ.get("AREAL_SP_OVERLAP", "1") == "1" and \
                    comm.overlap(g.group("tp")) is None:
                # the pipelined linears own a second exchange object: its calls sit on a side stream
                from ..parallel.sp_overlap import SPOverlap

                pc2 = peer_collectives.open_for(g.group("tp"), self.device)
                comm.attach_overlap(g.group("tp"), SPOverlap(pc2) if pc2 is not None else None)
        edp = max(1, len(g.group_ranks["edp"]))  # replicas of every expert shard
        flat = self.model.flatten(pad_to=g.dp * g.cp, expert_pad_to=edp if g.ep > 1 else None)
        if cfg.init_from_scratch or not (cfg.path and os.path.isdir(cfg.path)):
            # same seed on every dp replica; tp/pp shards draw from different streams
            self.model.init_weights(seed=1 + 1000 * g.coord.tp + 100000 * g.coord.pp)
            if g.pp > 1:
                from ..parallel.pipeline import sync_tied_embedding_weights

                sync_tied_embedding_weights(self)
            if g.tp > 1:  # parameters replicated over tp (norms, embeddings, router) must start identical
                self._replica_sync(flat.flat, self._tp_replicated_ranges(), "first")
            if self.is_critic and cfg.init_critic_from_actor:
                pass
        else:
            hf_io.load_hf_weights(self.model, cfg.path, strict=not (self.is_critic and cfg.init_critic_from_actor))
        with_optim = self.optimizer_config is not None and self.optimizer_config.type != "empty"
        if int(cfg.native.zero_stage) == 3 or cfg.fsdp.offload_params:
            from ..parallel.zero3 import Zero3

            assert g.pp == 1 and g.ep == 1 and self.model.flat_experts is None, \
                "zero_stage 3 shards over data / context parallel ranks (of every tensor-parallel rank); combine " \
                "pipeline or expert parallelism with zero_stage 1"
            rep = self._spans_by_name(self._tp_replicated_ranges()) if g.tp > 1 else []
            partial = self._spans_by_name(self._partial_grad_ranges() + (
                self._kv_ranges() if self._kv_replication() > 1 else [])) if g.tp > 1 else []
            self.zero3 = Zero3(self.model, self.optimizer_config if with_optim else None, group=g.group("dp_cp"),
                               reshard_after_forward=cfg.native.reshard_after_forward,
                               offload_optimizer=cfg.native.offload_optimizer or cfg.fsdp.offload_params,
                               tp_group=g.group("tp"), tp_replicated=rep, tp_partial_grads=partial)
            if with_optim:
                self.optimizer = self.zero3
                self._total_steps = max(1, ft_spec.total_train_steps if ft_spec is not None else 1)
        elif with_optim:
            flat.alloc_main_grad()
            mp_groups = tuple(x for x in (g.group("tp"), g.group("pp")) if x is not None)
            zpeer = (peer_collectives.open_for(g.group("dp_cp"), self.device)
                     if use_peer and g.dp * g.cp > 1 else None)
            self.optimizer = ShardedAdamW(flat, self.optimizer_config, dp_group=g.group("dp_cp"), mp_groups=mp_groups,
                                          replicated_ranges=self._norm_ranges(), peer=zpeer)
            fe = self.model.flat_experts
            if fe is not None:
                fe.alloc_main_grad()
                # groups whose members hold DIFFERENT expert parameters (their squared norms add up): whole
                # experts are not split over tp - those ranks are part of the ep / edp groups
                emp = tuple(x for x in (g.group("tp") if g.etp == g.tp else None, g.group("pp"), g.group("ep"))
                            if x is not None)
                self.expert_optimizer = ShardedAdamW(fe, self.optimizer_config, dp_group=g.group("edp"), mp_groups=emp)
            self._total_steps = max(1, ft_spec.total_train_steps if ft_spec is not None else 1)
        self.initialized = True
        logger.info(f"model {mc.family} ({flat.numel / 1e6:.1f}M params on this rank) ready in "
                    f"{time.perf_counter() - tik:.2f}s")

    def _spans_by_name(self, ranges: List[Tuple[int, int, int]]) -> List[Tuple[str, int, int, int]]:
        """Ranges of the flat parameter buffer as (parameter name, first element, end element, value)."""
        out = []
        for a, b, v in ranges:
            for n, (o, numel) in self.model.flat.offsets.items():
                if o <= a and b <= o + numel:
                    out.append((n, a - o, b - o, v))
                    break
            else:
                raise AssertionError(f"flat range [{a}, {b}) spans more than one parameter")
        return out

    def _zero3_tp_tag(self) -> str:
        g = self.grid
        return f"-tp{g.coord.tp}of{g.tp}" if g.tp > 1 else ""

    def _kv_replication(self) -> int:
        return max(1, self.grid.tp // self.model_config.n_kv_heads)

    def _kv_ranges(self) -> List[Tuple[int, int, int]]:
        """Flat ranges of the K/V rows of every fused qkv weight / bias (replicated `kv_rep` times)."""
        D = self.model_config.head_dim
        kv_rep = self._kv_replication()
        out = []
        for n, (o, numel) in self.model.flat.offsets.items():
            if n.split(".")[-1] in ("qkv", "qkv_bias"):
                attn = self.model.layers[0].attn
                per_row = numel // ((attn.hq + 2 * attn.hkv) * D)
                out.append((o + attn.hq * D * per_row, o + numel, kv_rep))
        return out

    def _tp_replicated_ranges(self) -> List[Tuple[int, int, int]]:
        """(lo, hi, replication) flat ranges of parameters that more than one tp rank holds."""
        rep = []
        names = {"ln1", "ln2", "norm", "q_norm", "k_norm", "router", "ln1_bias", "ln2_bias", "norm_bias", "pos_embed",
                 "o_bias", "proj_bias"}
        if not self.model.vocab_parallel:
            names |= {"embed", "head"}
        elif self.model_config.is_critic:
            names.add("head")  # the value head [1, H] is not a vocabulary tensor
        for n, (o, numel) in self.model.flat.offsets.items():
            if n.split(".")[-1] in names:
                rep.append((o, o + numel, self.grid.tp))
        if self._kv_replication() > 1:
            rep += self._kv_ranges()
        return sorted(rep)

    def _norm_ranges(self) -> List[Tuple[int, int, float]]:
        """Ranges whose squared-norm contribution must be down-weighted: tp replicas count once, and the
        last pipeline stage's copy of a tied embedding counts zero times (the first stage counts it)."""
        g = self.grid
        ranges = self._tp_replicated_ranges() if g.tp > 1 else []
        mc = self.model_config
        if g.pp > 1 and mc.tie_word_embeddings and not mc.is_critic and g.is_last_stage():
            o, numel = self.model.flat.offsets["head"]
            ranges = [r for r in ranges if r[0] != o] + [(o, o + numel, float("inf"))]
        return sorted(ranges)

    def _replica_sync(self, flat: torch.Tensor, ranges: List[Tuple[int, int, int]], op: str):
        """Make replicated ranges consistent inside every set of `rep` neighbouring tp ranks: `sum` adds
        the replicas' values (gradients), `first` copies the first replica's values (initial weights)."""
        g = self.grid
        if g.tp == 1 or not ranges:
            return
        for rep in sorted({r for _, _, r in ranges}):
            sel = [(a, b) for a, b, r in ranges if r == rep]
            packed = torch.cat([flat[a:b] for a, b in sel])
            allp = comm.all_gather_cat(packed.unsqueeze(0), g.group("tp"))
            s0 = (g.coord.tp // rep) * rep
            packed = allp[s0:s0 + rep].sum(0) if op == "sum" else allp[s0]
            o = 0
            for a, b in sel:
                flat[a:b].copy_(packed[o:o + (b - a)])
                o += b - a

    def _partial_grad_ranges(self) -> List[Tuple[int, int, int]]:
        """Replicated parameters whose gradient on one tp rank is only a partial sum: per-head q/k norm
        weights see only the local heads; under sequence parallelism the block norms and the final norm
        see only the local tokens (reference: ``realhf/impl/model/backend/megatron.py:227-259``)."""
        names = {"q_norm", "k_norm"}
        if self.model.seq_par:
            names |= {"ln1", "ln2", "norm"}
        return sorted((o, o + numel, self.grid.tp) for n, (o, numel) in self.model.flat.offsets.items()
                      if n.split(".")[-1] in names)

    def _sync_replicated_kv_grads(self):
        """When tp exceeds the KV heads, `kv_rep` neighbouring ranks hold the same K/V projection but each
        sees only the gradient that flows through its own query heads: sum them inside every replica set.
        Partial gradients of other replicated parameters are summed over the whole tp group."""
        if self.grid.tp > 1 and self._kv_replication() > 1:
            self._replica_sync(self.model.flat.main_grad, self._kv_ranges(), "sum")
        if self.grid.tp > 1:
            self._replica_sync(self.model.flat.main_grad, self._partial_grad_ranges(), "sum")

    def destroy(self):
        if getattr(self, "_ipc_export", None) is not None:
            self._ipc_export.close()
            self._ipc_export = None
        self.model = None
        self.optimizer = None
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        if self.own_global_group and dist.is_initialized():
            dist.destroy_process_group()
        self.initialized = False

    def train(self, mode: bool = True):
        self.onload()  # every compute entry point passes through here: a parked model comes back first
        self._training = mode
        if self.model is not None:
            self.model.train(mode)
        return self

    def set_version(self, version: int):
        self._version = version

    def get_version(self) -> int:
        return self._version

    # ------------------------------------------------------------------------ LR schedule
    def current_lr(self) -> float:
        oc = self.optimizer_config
        return oc.lr * lr_lambda(self._lr_step, self._total_steps, oc)

    def step_lr_scheduler(self):
        self._lr_step += 1

    # ------------------------------------------------------------------- micro-batching
    def prepare_mb_list(self, input_: Dict[str, Any]) -> MicroBatchList:
        assert "attention_mask" in input_ and "input_ids" in input_
        input_ = amend_position_ids(dict(input_))
        mb_list = split_padded_tensor_dict_into_mb_list(input_, self.config.mb_spec, group=self.grid.group("dp"))
        align = 1
        if self.grid.cp > 1:
            align = getattr(self.model.sp, "align", self.grid.cp)
        if self.grid.tp > 1 and self.config.native.sequence_parallel:
            align = math.lcm(align, self.grid.tp)
        mb_list = pad_mb_list(mb_list, pad_value=0.0, pad_to_maximum=self.config.pad_to_maximum, align_to=align)
        return mb_list.to(self.device)

    def _run_model(self, mb: Dict[str, Any], hidden_in=None, residual_in=None, chunk: Optional[int] = None):
        """Packed (padded) micro-batch -> final hidden states [T_padded (/cp), H]; on a pipeline stage
        that is not the last, the (h, residual) pair handed to the next stage. `chunk`: the layer slice of an
        interleaved pipeline stage to run."""
        ids = mb["input_ids"]
        pos = mb["position_ids"].to(torch.int32)
        cu = mb["cu_seqlens"].to(torch.int32)
        sp = self.model.sp
        if self.model_config.vision is not None and mb.get("multi_modal_input") is not None:
            assert sp is None, "vision-language batches do not combine with context parallelism"
            assert chunk is None, "vision-language batches run on contiguous pipeline stages"
            return self._run_multimodal(mb, ids, cu, hidden_in, residual_in)
        if sp is not None:
            ids, pos = sp.local_slice(ids), sp.local_slice(pos)
        return self.model(ids, pos, cu, int(mb["max_seqlen"]), hidden_in, residual_in, chunk=chunk)

    def _run_multimodal(self, mb: Dict[str, Any], ids: torch.Tensor, cu: torch.Tensor, hidden_in, residual_in):
        """Packed vision-language micro-batch: `multi_modal_input[i]` holds the pixel patches and grids of the
        images of sequence i. Image placeholders take the vision tower's outputs; every sequence gets its
        own 3-D rotary positions (the appended padding sequence, which has no entry, counts as text)."""
        from ..models.vision import mrope_position_ids

        mc = self.model_config
        mm = list(mb["multi_modal_input"])
        bounds = cu.tolist()
        host_ids = ids.tolist()
        img_tokens = [t for t in (mc.image_token_id, mc.video_token_id) if t is not None]
        merge = int(mc.vision.get("spatial_merge_size", 2))
        pos3, masks, pixels, grids = [], [], [], []
        for i in range(len(bounds) - 1):
            item = mm[i] if i < len(mm) else None
            g = item["image_grid_thw"] if item is not None and item.get("image_grid_thw") is not None else None
            g_list = g.tolist() if g is not None else []
            p3, _, msk = mrope_position_ids(host_ids[bounds[i]:bounds[i + 1]], g_list, img_tokens, merge, True)
            pos3.append(p3)
            masks.append(msk)
            if g is not None and len(g_list) > 0:
                pixels.append(item["pixel_values"])
                grids.append(g)
        positions = torch.cat(pos3, 1).to(ids.device)
        if self.model.has_embed and hidden_in is None:
            px = torch.cat(pixels, 0) if pixels else None
            hidden_in = self.model.embed_multimodal(ids, px, torch.cat(grids, 0) if grids else None,
                                                    torch.cat(masks))
        return self.model(None, positions, cu, int(mb["max_seqlen"]), hidden_in, residual_in)

    def _proxy(self, hidden: torch.Tensor, pad_len: int) -> LogitsProxy:
        nc = self.config.native
        return LogitsProxy(self.model, hidden, nc.logprob_chunk, nc.keep_logits, self.grid.group("tp"),
                           sp=self.model.sp, pad_len=pad_len)

    @staticmethod
    def _unpadded(mb_padded: Dict[str, Any], pad_len: int, old_cu: torch.Tensor) -> Dict[str, Any]:
        if pad_len == 0:
            return mb_padded
        total = int(old_cu[-1])
        out = {}
        for k, v in mb_padded.items():
            if k == "cu_seqlens":
                out[k] = old_cu
            elif k == "max_seqlen":
                lens = old_cu[1:] - old_cu[:-1]
                out[k] = int(lens.max()) if lens.numel() else 0
            elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == total + pad_len:
                out[k] = v[:total]
            else:
                out[k] = v
        return out

    # ------------------------------------------------------------------------ train_batch
    def train_batch(self, input_: Dict[str, Any], loss_fn: Callable, loss_weight_fn: Callable) -> Dict[str, float]:
        assert self.optimizer is not None, "engine was initialised without an optimizer"
        self.train(True)
        if self.grid.pp > 1:
            from ..parallel.pipeline import pipeline_train_batch

            return pipeline_train_batch(self, input_, loss_fn, loss_weight_fn)
        self.zero_main_grads()
        mb_list = self.prepare_mb_list(input_)
        weights = [loss_weight_fn(mb) for mb in mb_list.mbs]
        total_w = torch.stack([torch.as_tensor(w, dtype=torch.float32, device=self.device) for w in weights]).sum()
        comm.all_reduce_sum(total_w, self.grid.group("dp"))
        assert float(total_w) > 0, "total loss weight must be positive"
        for i, (pmb, pad_len, old_cu) in enumerate(zip(mb_list.padded_mbs, mb_list.padding_lengths,
                                                       mb_list.old_cu_seqlens_list)):
            self._begin_mb()
            hidden = self._run_model(pmb)
            proxy = self._proxy(hidden, pad_len)
            mb = self._unpadded(pmb, pad_len, old_cu)
            loss = loss_fn(proxy, mb)
            scale = torch.as_tensor(weights[i], dtype=torch.float32, device=self.device) / total_w
            aux = self.model.moe_aux_loss()
            if aux is not None:
                loss = loss + aux
            if self.zero3 is not None:
                self.zero3.in_backward = True
            elif i == len(mb_list.padded_mbs) - 1 and self._can_overlap_grad_reduce():
                self.optimizer.arm_overlap()
            (loss * scale).backward()
            self._end_mb()
        return self.optimizer_step()

    def _can_overlap_grad_reduce(self) -> bool:
        """Gradient buckets may leave during the backward only if nothing still has to touch the full buffer
        afterwards: tensor-parallel replicas whose gradients are partial sums are combined on the full buffer
        first, and routed experts have an optimizer (and a group) of their own."""
        g = self.grid
        if not self.config.native.overlap_grad_reduce or self.expert_optimizer is not None:
            return False
        return not (g.tp > 1 and (self._kv_replication() > 1 or self._partial_grad_ranges()))

    def _begin_mb(self):
        """ZeRO-3: the parameters outside the decoder blocks become whole for this micro-batch. Dropout masks move
        on to the next micro-batch's keys."""
        from ..utils import rng

        rng.keys.advance()
        if self.zero3 is not None:
            self.zero3.begin_microbatch()

    def _end_mb(self):
        if self.zero3 is not None:
            self.zero3.end_microbatch()

    def full_params(self):
        """Context in which every parameter of this rank's model is whole (no-op unless ZeRO-3 shards them)."""
        import contextlib

        return self.zero3.full_params() if self.zero3 is not None else contextlib.nullcontext()

    def zero_main_grads(self):
        if self.zero3 is not None:
            return self.zero3.zero_grad()
        self.model.flat.main_grad.zero_()
        if self.model.flat_experts is not None:
            self.model.flat_experts.main_grad.zero_()

    def optimizer_step(self) -> Dict[str, float]:
        """Reduce gradients, clip by the global norm (dense + expert parameters), AdamW, re-gather params."""
        lr = self.current_lr()
        if self.zero3 is None:  # ZeRO-3 sums the partial gradients on its shards, inside its step
            self._sync_replicated_kv_grads()
        if self.expert_optimizer is None:
            self.optimizer.step(lr=lr)
        else:
            d, e = self.optimizer.prepare(), self.expert_optimizer.prepare()
            total = d + e
            d.copy_(total)
            e.copy_(total)
            self.optimizer.apply(lr)
            self.expert_optimizer.apply(lr)
        if hasattr(self.optimizer, "step_stats"):
            norm, ok = self.optimizer.step_stats()
        else:
            norm, ok = self.optimizer.grad_norm(), self.optimizer.update_successful()
        return dict(update_successful=float(ok), grad_norm=norm, lr=lr)

    @torch.no_grad()
    def eval_batch(self, input_: Dict[str, Any], loss_fn: Callable, loss_weight_fn: Callable) -> Optional[torch.Tensor]:
        self.train(False)
        if self.grid.pp > 1:
            from ..parallel.pipeline import pipeline_eval_batch

            return pipeline_eval_batch(self, input_, loss_fn, loss_weight_fn)
        mb_list = self.prepare_mb_list(input_)
        weights = [loss_weight_fn(mb) for mb in mb_list.mbs]
        total_w = torch.stack([torch.as_tensor(w, dtype=torch.float32, device=self.device) for w in weights]).sum()
        comm.all_reduce_sum(total_w, self.grid.group("dp"))
        total = torch.zeros((), dtype=torch.float32, device=self.device)
        for i, (pmb, pad_len, old_cu) in enumerate(zip(mb_list.padded_mbs, mb_list.padding_lengths,
                                                       mb_list.old_cu_seqlens_list)):
            self._begin_mb()
            proxy = self._proxy(self._run_model(pmb), pad_len)
            loss = loss_fn(proxy, self._unpadded(pmb, pad_len, old_cu))
            total += loss.float() * torch.as_tensor(weights[i], dtype=torch.float32, device=self.device) / total_w
            self._end_mb()
        comm.all_reduce_sum(total, self.grid.group("dp"))
        return total

    @torch.no_grad()
    def forward(self, input_: Dict[str, Any], output_seqlens: Optional[List[int]] = None,
                post_hook: Optional[Callable] = None, aggregate_fn: Callable = torch.cat) -> Any:
        """No-grad forward over micro-batches. `post_hook(proxy, mb)` maps the LM-head proxy to a per-token
        tensor; results are re-ordered to the input order and re-padded to [B, S]."""
        self.train(False)
        if self.grid.pp > 1:
            from ..parallel.pipeline import pipeline_forward

            return pipeline_forward(self, input_, output_seqlens, post_hook, aggregate_fn)
        B, S = input_["attention_mask"].shape
        mb_list = self.prepare_mb_list(input_)
        if output_seqlens is None:
            output_seqlens = input_["attention_mask"].sum(-1).long().tolist()
        results = []
        for pmb, pad_len, old_cu in zip(mb_list.padded_mbs, mb_list.padding_lengths, mb_list.old_cu_seqlens_list):
            self._begin_mb()
            proxy = self._proxy(self._run_model(pmb), pad_len)
            mb = self._unpadded(pmb, pad_len, old_cu)
            out = post_hook(proxy, mb) if post_hook is not None else proxy.materialize()
            results.append(out)
            self._end_mb()
        res = aggregate_fn(results)
        lens_fwd = [output_seqlens[i] for i in mb_list.forward_indices]
        seqs = unpack_sequence(res, lens=lens_fwd)
        ordered = [seqs[i] for i in mb_list.backward_indices]
        out = res.new_zeros(B, S, *res.shape[1:])
        for b, s in enumerate(ordered):
            out[b, : s.shape[0]] = s
        return out

    @torch.no_grad()
    def generate(self, prompts: List[List[int]], gconfig=None, n_micro_batches: Optional[int] = None, **kw):
        """Generate with the weights this trainer holds, through its own pipeline stages and tensor-parallel
        ranks (``parallel/pipeline_generate.py``). Every rank of a model-parallel replica passes the same
        prompts and receives the same list of sequences; data-parallel replicas generate for their own prompts.
        `gconfig`: a ``GenerationHyperparameters``; keyword arguments override its fields. Reference:
        ``realhf/impl/model/backend/pipe_runner.py:420-646`` (pipelined generation of the legacy trainer)."""
        from ..parallel.pipeline_generate import PipelineGenerator

        assert self.zero3 is None and self.grid.cp == 1, "generation through the trainer needs resident, " \
            "unsharded-by-context weights (ZeRO-1, cp = 1)"
        self.train(False)
        if getattr(self, "_generator", None) is None:
            self._generator = PipelineGenerator(self.model, self.grid, self.device,
                                                seed=1 + 7919 * self.data_parallel_rank)
        args = {}
        if gconfig is not None:
            args = dict(max_new_tokens=gconfig.max_new_tokens, min_new_tokens=gconfig.min_new_tokens,
                        temperature=gconfig.temperature, top_p=gconfig.top_p, top_k=gconfig.top_k,
                        greedy=gconfig.greedy, stop_token_ids=list(gconfig.stop_token_ids or []))
        args.update(kw)
        return self._generator.generate(prompts, n_micro_batches=n_micro_batches, **args)

    # ------------------------------------------------------------------- offload / onload
    def offload(self) -> int:
        """Park this model in pinned host memory while it is idle: the flat parameter buffer(s), the optimizer's
        fp32 master / moment shards (copied out) and the fp32 gradient buffer(s) (dropped: zeroed before their
        next use anyway). Returns the device bytes freed. Any later call that needs the model brings it back
        (`onload`). Role of the reference's offload hook (``realhf/system/model_worker.py``, ``OffloadHook`` in
        ``realhf/api/core/dfg.py``)."""
        from ..utils.offload import ParkedStorages

        if self.zero3 is not None:
            return 0  # ZeRO-3 keeps shards only and has its own host-resident optimizer option
        if getattr(self, "_parked", None) is None:
            self._parked = ParkedStorages()
        if self._parked.parked:
            return 0
        tensors, keep = [], []
        for fp in (self.model.flat, self.model.flat_experts):
            if fp is None:
                continue
            tensors += [fp.flat, fp.main_grad]
            keep += [True, False]
        for opt in (self.optimizer, self.expert_optimizer):
            if opt is None:
                continue
            tensors += [opt.master, opt.m, opt.v]
            keep += [True, True, True]
        freed = self._parked.park(tensors, keep)
        if self.device.type == "cuda":
            torch.cuda.empty_cache()
        logger.info(f"offloaded {freed / 2**30:.2f} GiB to host memory")
        return freed

    def onload(self) -> int:
        parked = getattr(self, "_parked", None)
        if parked is None or not parked.parked:
            return 0
        return parked.unpark()

    @property
    def offloaded(self) -> bool:
        parked = getattr(self, "_parked", None)
        return parked is not None and parked.parked

    # ----------------------------------------------------------------------- save / load
    def _save_sharded(self, meta: SaveLoadMeta):
        """Distributed checkpoint: every (pp, tp) shard writes its own slice of the flat buffer - no
        gather, no conversion - plus a manifest with the parameter offsets; dp replica 0 writes the dense
        buffer, and with expert parallelism every rank of the first expert-data-parallel replica writes the
        routed experts IT holds under its expert-parallel rank. The optimizer state is per rank as in the HF
        format. Role of the reference's ``dcp`` / ``naive_distributed`` formats
        (``areal/experimental/utils/megatron_checkpointer.py``, ``areal/experimental/autotp_engine.py:97``)."""
        import json

        from safetensors.torch import save_file

        g = self.grid
        os.makedirs(meta.path, exist_ok=True)
        if self.zero3 is not None:  # every data-parallel rank owns a different slice of every parameter
            z = self.zero3
            save_file({"param_shard": z.param_shard.detach().cpu()},
                      os.path.join(meta.path, f"zero3-shard-{z.rank}-of-{z.n}{self._zero3_tp_tag()}.safetensors"))
            if z.rank == 0 and g.coord.tp == 0:
                with open(os.path.join(meta.path, "config.json"), "w") as f:
                    json.dump(self.model_config.to_hf_dict(), f, indent=2)
            if meta.with_optim and self.optimizer is not None:
                torch.save(dict(optim=z.state_dict(), lr_step=self._lr_step),
                           os.path.join(meta.path, f"optim_world_size_{self.world_size}_rank_{self.rank}.pt"))
            if dist.is_initialized():
                dist.barrier(group=self.grid.group("world"))
            return
        tag = f"pp{g.coord.pp}-tp{g.coord.tp}"
        if g.coord.dp == 0 and g.coord.cp == 0:
            save_file({"flat": self.model.flat.flat.detach().cpu()}, os.path.join(meta.path, f"shard-{tag}.safetensors"))
            manifest = dict(format="areal-sharded-v2", layout=dict(dp=g.dp, tp=g.tp, pp=g.pp, cp=g.cp, ep=g.ep, etp=g.etp, vpp=g.vpp),
                            dtype=str(self.model.dtype).replace("torch.", ""),
                            offsets={n: [int(o), int(k)] for n, (o, k) in self.model.flat.offsets.items()},
                            layer_range=list(self.model.layer_range))
            with open(os.path.join(meta.path, f"manifest-{tag}.json"), "w") as f:
                json.dump(manifest, f)
            if g.coord.pp == 0 and g.coord.tp == 0:
                with open(os.path.join(meta.path, "config.json"), "w") as f:
                    json.dump(self.model_config.to_hf_dict(), f, indent=2)
        if self.model.flat_experts is not None and g.holds_first_expert_replica():
            save_file({"flat_experts": self.model.flat_experts.flat.detach().cpu()},
                      os.path.join(meta.path, self._expert_shard_file()))
        if meta.with_optim and self.optimizer is not None:
            sd = dict(optim=self.optimizer.state_dict(), lr_step=self._lr_step)
            if self.expert_optimizer is not None:
                sd["expert_optim"] = self.expert_optimizer.state_dict()
            torch.save(sd, os.path.join(meta.path, f"optim_world_size_{self.world_size}_rank_{self.rank}.pt"))
        if dist.is_initialized():
            dist.barrier(group=self.grid.group("world"))

    def _load_sharded(self, meta: SaveLoadMeta):
        """Load a sharded checkpoint. Written with the same layout it is one file read per rank (two with expert
        parallelism); written with ANOTHER tensor / pipeline / expert-parallel layout, every rank pulls the intervals
        of the source shard files that make up its own flat buffers (the piece algebra of
        ``parallel/param_realloc.py``, applied to files instead of peers; dense and mixture-of-experts models): the
        resharding load of ``areal/experimental/utils/megatron_checkpointer.py`` /
        ``areal/experimental/model/hf_load.py``."""
        import glob
        import json

        import numpy as np
        from safetensors.torch import load_file

        g = self.grid
        if self.zero3 is not None:
            z = self.zero3
            fn = os.path.join(meta.path, f"zero3-shard-{z.rank}-of-{z.n}{self._zero3_tp_tag()}.safetensors")
            if not os.path.exists(fn):
                raise FileNotFoundError(f"{fn}: a ZeRO-3 sharded checkpoint is read back by the same number of "
                                        "data-parallel ranks (convert through the hf format otherwise)")
            t = load_file(fn)["param_shard"]
            assert t.numel() == z.param_shard.numel(), "parameter shard layout of the checkpoint differs"
            z.param_shard.copy_(t.to(z.param_shard.dtype))
            return
        mans = sorted(glob.glob(os.path.join(meta.path, "manifest-pp*-tp*.json")))
        if not mans:
            raise FileNotFoundError(f"no sharded checkpoint manifest under {meta.path}")
        with open(mans[0]) as f:
            lay = json.load(f)["layout"]
        flat = self.model.flat.flat
        moe = self.model_config.n_experts > 0
        same = (lay["tp"], lay["pp"], int(lay.get("vpp", 1))) == (g.tp, g.pp, g.vpp)
        if moe:
            same = same and (int(lay.get("ep", 1)), int(lay.get("etp", lay["tp"]))) == (g.ep, g.etp)
        if same:
            tag = f"pp{g.coord.pp}-tp{g.coord.tp}"
            with open(os.path.join(meta.path, f"manifest-{tag}.json")) as f:
                man = json.load(f)
            mine = {n: [int(o), int(k)] for n, (o, k) in self.model.flat.offsets.items()}
            assert man["offsets"] == mine, "parameter layout of the checkpoint differs from this model"
            t = load_file(os.path.join(meta.path, f"shard-{tag}.safetensors"))
            flat.copy_(t["flat"].to(flat.dtype))
            if self.model.flat_experts is not None:
                fe = self.model.flat_experts.flat
                t = load_file(os.path.join(meta.path, self._expert_shard_file()))
                fe.copy_(t["flat_experts"].to(fe.dtype))
            return
        from ..parallel.param_realloc import derive_file_load_plan

        bufs = {"flat": self.model.flat}
        if self.model.flat_experts is not None:
            bufs["flat_experts"] = self.model.flat_experts
        host = {k: torch.zeros(fp.flat.numel(), dtype=fp.flat.dtype) for k, fp in bufs.items()}
        filled = {k: 0 for k in bufs}
        for fname, key, which, tri in derive_file_load_plan(self.model_config, lay, self.model):
            sflat = load_file(os.path.join(meta.path, fname))[key].to(host[which].dtype)
            ops.set_intervals(ops.slice_intervals(sflat, np.stack([tri[:, 0], tri[:, 0] + tri[:, 2]], 1)), host[which],
                              np.stack([tri[:, 1], tri[:, 1] + tri[:, 2]], 1))
            filled[which] += int(tri[:, 2].sum())
        for k, fp in bufs.items():
            used = sum(n for _, n in fp.offsets.values())
            assert filled[k] == used, f"resharding load covered {filled[k]} of {used} elements of {k}"
            fp.flat.copy_(host[k])
        logger.info(f"resharded checkpoint load: tp{lay['tp']} pp{lay['pp']} vpp{lay.get('vpp', 1)} ep{lay.get('ep', 1)} "
                    f"-> tp{g.tp} pp{g.pp} vpp{g.vpp} ep{g.ep}")

    def _expert_shard_file(self) -> str:
        g = self.grid
        return f"experts-pp{g.coord.pp}-tp{g.coord.tp if g.etp == g.tp else 0}-ep{g.rank_in('ep')}.safetensors"

    def save(self, meta: SaveLoadMeta):
        self.onload()
        if meta.weight_format in ("dcp", "naive_distributed"):
            return self._save_sharded(meta)
        if meta.weight_format != "hf":
            raise NotImplementedError(f"weight format {meta.weight_format} (use 'hf', 'dcp' or 'naive_distributed')")
        g = self.grid
        # one writer per pipeline stage: dp rank 0, cp rank 0, tp rank 0 (tp shards are gathered)
        # (with expert parallelism every ep rank of the first expert-data-parallel replica also writes
        # its own experts)
        if g.etp == g.tp:
            participate = g.coord.dp < g.ep and g.coord.cp == 0
            writer = participate and g.coord.tp == 0
            experts_only = g.coord.dp > 0
        else:
            # whole experts: stage ranks 0 .. ep-1 (tp fastest) hold the distinct expert sets; the dense tensors
            # need every rank of a participating tp group for their gathers, rank 0 writes them
            participate = g.coord.dp < max(1, g.ep // g.tp)
            writer = g.holds_first_expert_replica()
            experts_only = g.expert_slot() > 0
        tag = f"-stage{g.coord.pp}" if g.pp > 1 else ""
        if g.ep > 1:
            tag += f"-ep{g.rank_in('ep')}"
        wm = {}
        with self.full_params():  # ZeRO-3: every rank helps to make the writer's parameters whole
            if participate:
                wm = hf_io.save_hf_checkpoint(self.model, meta.path, stage_tag=tag, write=writer,
                                              experts_only=experts_only)
        if dist.is_initialized() and (g.pp > 1 or g.ep > 1):
            maps = [None] * g.world
            dist.all_gather_object(maps, wm if writer else {}, group=g.group("world"))
        else:
            maps = [wm]
        if g.global_rank == g.ranks[g.index_of(g.pp - 1, 0, 0, 0)]:
            hf_io.finalize_hf_checkpoint(meta.path, self.model_config, maps, meta.tokenizer)
            if meta.processor is not None:
                meta.processor.save_pretrained(meta.path)
        if meta.with_optim and self.optimizer is not None:
            os.makedirs(meta.path, exist_ok=True)
            sd = dict(optim=self.optimizer.state_dict(), lr_step=self._lr_step)
            if self.expert_optimizer is not None:
                sd["expert_optim"] = self.expert_optimizer.state_dict()
            torch.save(sd, os.path.join(meta.path, f"optim_world_size_{self.world_size}_rank_{self.rank}.pt"))
        if dist.is_initialized():
            dist.barrier(group=self.grid.group("world"))

    def load(self, meta: SaveLoadMeta):
        self.onload()
        if meta.weight_format in ("dcp", "naive_distributed"):
            self._load_sharded(meta)
        elif meta.weight_format == "hf":
            with self.full_params():
                hf_io.load_hf_weights(self.model, meta.path)
                if self.zero3 is not None:
                    self.zero3.reshard_from_full()
        else:
            raise NotImplementedError(f"weight format {meta.weight_format} (use 'hf', 'dcp' or 'naive_distributed')")
        if self.optimizer is not None:
            self.optimizer.resync_master_from_params()
            if self.expert_optimizer is not None:
                self.expert_optimizer.resync_master_from_params()
            if meta.with_optim:
                p = os.path.join(meta.path, f"optim_world_size_{self.world_size}_rank_{self.rank}.pt")
                sd = torch.load(p, map_location="cpu", weights_only=True)
                self.optimizer.load_state_dict(sd["optim"])
                if self.expert_optimizer is not None:
                    self.expert_optimizer.load_state_dict(sd["expert_optim"])
                self._lr_step = int(sd["lr_step"])
        if dist.is_initialized():
            dist.barrier(group=self.grid.group("world"))

    # ---------------------------------------------------------------------- weight upload
    def get_param_specs(self, weight_chunked_mem_mb: int = 1024) -> List[List[ParamSpec]]:
        """HF-named parameter specs grouped (FFD) into chunks of about `weight_chunked_mem_mb`."""
        specs = [ParamSpec(name=n, shape=tuple(t.shape), dtype=str(t.dtype).replace("torch.", ""))
                 for n, t in self._iter_full_hf_params(meta_only=True)]
        sizes = [s.size for s in specs]
        cap = max(max(sizes), weight_chunked_mem_mb * 1024 * 1024)
        groups = datapack.ffd_allocate(sizes, cap, 1)
        return [[specs[i] for i in sorted(g)] for g in sorted(groups, key=lambda g: min(g))]

    def _iter_full_hf_params(self, meta_only: bool = False):
        if meta_only and self.zero3 is not None:  # shapes only, no collective: an unsharded copy on the meta device
            twin = Transformer(self.model_config, dtype=self.model.dtype, device="meta")
            yield from hf_io.native_to_hf_state(twin, gather=False)
        elif meta_only and self.grid.tp == 1:
            yield from hf_io.native_to_hf_state(self.model, gather=False)
        else:
            yield from hf_io.native_to_hf_state(self.model, gather=True)

    def share_pipeline_stats(self, stats: Dict[str, float]) -> Dict[str, float]:
        """Loss functions run on the last pipeline stage only, so their statistics (actor loss, entropy, clip
        ratios ...) are recorded there, while the logging sink is global rank 0 = stage 0. Every stage gets
        the union; values this stage exported itself win. No-op without pipeline parallelism."""
        if self.grid.pp <= 1 or not dist.is_initialized():
            return stats
        group = self.grid.group("pp")
        ranks = dist.get_process_group_ranks(group)
        box = [stats if self.grid.coord.pp == self.grid.pp - 1 else None]
        dist.broadcast_object_list(box, src=ranks[self.grid.pp - 1], group=group)
        return {**(box[0] or {}), **stats}

    def connect_local_engine(self, engine):
        """Register a rollout engine living in this process (colocated mode)."""
        self._local_rollout_engine = engine

    def upload_weights(self, meta: WeightUpdateMeta):
        self.onload()
        if meta.type == "inproc":
            # Colocated: the data-parallel head of every replica hosts a rollout engine with the FULL model.
            eng = self._local_rollout_engine
            if self.grid.pp > 1:
                raise NotImplementedError("a colocated rollout engine needs a trainer without pipeline stages "
                                          "(use a decoupled allocation such as native:d4+native:d2p2)")
            if self.grid.tp == 1 and self.zero3 is None:
                if eng is not None:  # same flat layout: one device copy
                    eng.load_weights_from_trainer(self.model, version=None,
                                                  abort_running=getattr(eng, "abort_on_weight_update", True))
            else:  # sharded trainer (tp, ZeRO-3): gather the shards (every rank takes part), load by name on the head
                with self.full_params():
                    full = dict(self._iter_full_hf_params())
                    if self.device.type == "cuda":
                        torch.cuda.current_stream().synchronize()  # the engine reads the tensors on its stream
                    if eng is not None:
                        eng.load_named_tensors(full, version=None,
                                               abort_running=getattr(eng, "abort_on_weight_update", True))
        elif meta.type == "disk":
            self._upload_disk(meta)
        elif meta.type in ("nccl", "xgmi"):
            self._upload_collective(meta)
        elif meta.type == "ipc":
            self._upload_ipc(meta)
        else:
            raise NotImplementedError(f"Unknown weight update type {meta.type}")

    def export_weight_buffers(self) -> Dict[str, Any]:
        """Export the memory behind this rank's bf16 parameters for peer-memory weight updates and gather the
        descriptors of ALL trainer ranks (collective over the world group). Every rank exports exactly what it holds -
        its flat buffer (a tensor-parallel slice of a pipeline stage; the whole model for a plain data-parallel
        replica) or, under ZeRO-3, its shard of that buffer - together with the layout the servers need to work out
        which byte ranges of which rank make up their own parameters (``parallel/param_realloc.py::derive_pull_plan``):
        every trainer GPU serves its own shard over its own links, nothing is gathered on the trainer
        (replaces the funnel of ``areal/engine/fsdp_engine.py:399-433``)."""
        from ..ops import ipc

        g = self.grid
        holds = self.zero3.param_shard if self.zero3 is not None else self.model.flat.flat
        if getattr(self, "_ipc_export", None) is None:
            self._ipc_export = ipc.export_buffer(holds)
            # expert parallelism: the routed experts this rank holds live in a buffer of their own (they differ
            # over the data-parallel ranks of a stage) and are exported next to the dense parameters
            fe = self.model.flat_experts
            self._ipc_export_experts = ipc.export_buffer(fe.flat) if fe is not None else None
        ex = self._ipc_export_experts
        mine = (self._ipc_export.desc, ex.desc if ex is not None else None)
        pairs = [mine]
        if dist.is_initialized() and self.world_size > 1:
            pairs = [None] * self.world_size
            dist.all_gather_object(pairs, mine, group=g.group("world"))
        descs = [p[0] for p in pairs]
        name, off, shape = self.model.flat.specs[-1]
        used = (off + int(math.prod(shape))) * self.model.flat.flat.element_size()
        replicas = g.dp * g.cp  # ranks that hold the same (pp, tp) slice; ZeRO-3 cuts the slice over them instead
        src = dict(dp=replicas, tp=g.tp, pp=g.pp, vpp=g.vpp, vocab_parallel=bool(self.model.vocab_parallel),
                   zero3=replicas if self.zero3 is not None else 1)
        if ex is not None:
            src.update(ep=g.ep, etp=g.etp, cp=g.cp, expert_buffers=[p[1] for p in pairs])
        return dict(ipc_buffers=descs, ipc_layout=flat_layout_fingerprint(self.model), ipc_used_bytes=int(used),
                    ipc_src=src)

    def _upload_ipc(self, meta: WeightUpdateMeta):
        """Peers pull; this side takes a snapshot of the flat buffer into the exported area (after the optimizer
        step's stream work; the trainer may then go on while the servers read) and says so: rank 0 posts the
        update key the client side waits for once every rank is ready."""
        self._ipc_export.refresh()
        if self._ipc_export_experts is not None:
            self._ipc_export_experts.refresh()
        if dist.is_initialized():
            dist.barrier(group=self.grid.group("world"))
        if self.rank == 0:
            key = names.update_weights_from_ipc(self.config.experiment_name, self.config.trial_name, self._version)
            name_resolve.add(key, str(time.time_ns()), keepalive_ttl=120, replace=True)

    def _upload_disk(self, meta: WeightUpdateMeta):
        self.save(SaveLoadMeta(path=meta.path, weight_format="hf", with_optim=False))
        if self.rank == 0:
            key = names.update_weights_from_disk(self.config.experiment_name, self.config.trial_name, self._version)
            name_resolve.add(key, str(time.time_ns()), keepalive_ttl=120, replace=True)
        if dist.is_initialized():
            dist.barrier(group=self.grid.group("world"))

    def _init_weight_update_group(self, meta: WeightUpdateMeta):
        from ..utils.distributed import init_custom_process_group

        n = max(1, meta.nccl_n_senders)
        assert self.grid.pp == 1, ("a pipeline stage holds only its own layers, the collective weight push would be "
                                   "partial: use the disk transport with pipeline-parallel trainers")
        assert n <= self.world_size, "more senders than trainer ranks"
        if self.rank < n and self._weight_update_group is None:
            ws = meta.alloc_mode.gen.world_size + 1
            self._weight_update_group = init_custom_process_group(
                backend="nccl" if self.device.type == "cuda" else "gloo",
                init_method=f"tcp://{meta.nccl_master_address}:{meta.sender_port(self.rank)}", world_size=ws, rank=0,
                group_name=meta.sender_group_name(self.rank))

    def _upload_collective(self, meta: WeightUpdateMeta):
        """Trainer ranks 0 .. n_senders-1 broadcast the (HF-named, full) parameters into their rollout groups,
        chunk i by rank i % n_senders, in the order of `meta.nccl_param_specs` (n_senders = 1 is the protocol of
        ``fsdp_engine.py:399-433``). The chunks of one round (one per sender) travel concurrently."""
        self._init_weight_update_group(meta)
        n = max(1, meta.nccl_n_senders)
        if self.zero3 is not None:
            with self.zero3.full_params():
                return self._broadcast_chunks(meta, n)
        return self._broadcast_chunks(meta, n)

    def _broadcast_chunks(self, meta: WeightUpdateMeta, n: int):
        # Only the tensors of this rank's own chunks are kept: a tensor-parallel rank still takes part in every
        # gather of the stream, but a full gathered copy of the model never exists on any rank, and ranks that
        # send nothing from an unsharded replica skip the stream altogether.
        mine_names = {spec.name for ci, chunk in enumerate(meta.nccl_param_specs) if ci % n == self.rank and self.rank < n
                      for spec in chunk}
        full = {}
        if mine_names or self.grid.tp > 1:
            for name, t in self._iter_full_hf_params():
                if name in mine_names:
                    full[name] = t
        for lo in range(0, len(meta.nccl_param_specs), n):
            mine = lo + self.rank
            if self.rank < n and mine < len(meta.nccl_param_specs):
                for spec in meta.nccl_param_specs[mine]:
                    t = full[spec.name].contiguous()
                    dist.broadcast(t, src=0, group=self._weight_update_group)
            if dist.is_initialized():
                dist.barrier(group=self.grid.group("world"))
        if self.device.type == "cuda":
            torch.cuda.synchronize()


def flat_layout_fingerprint(model: Transformer) -> str:
    """Identifies a flat parameter layout (names, offsets, shapes, dtype): two buffers with equal fingerprints
    can be copied byte for byte."""
    import hashlib

    text = repr([(n, int(o), tuple(int(x) for x in sh)) for n, o, sh in model.flat.specs]) + str(model.flat.flat.dtype)
    return hashlib.sha1(text.encode()).hexdigest()


# Names used by scripts written against the reference API.
FSDPEngine = NativeTrainEngine
MegatronEngine = NativeTrainEngine
AutoTPEngine = NativeTrainEngine  # tensor parallelism comes from the strategy's `t` dimension; no DeepSpeed
