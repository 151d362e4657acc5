"""Parameter re-allocation between two parallel layouts of the same model.

A model trained as (dp, tp, pp) can hand its weights to a copy laid out as (dp', tp', pp') - e.g. a
tensor-parallel trainer feeding data-parallel generation replicas - without going through a full
checkpoint: every destination shard pulls exactly the slices it needs from the source shards that own
them. Capability of ``realhf/impl/model/comm/param_realloc.py:73-563`` +
``realhf/impl/model/nn/real_llm_api.py:569-807`` (interval-based resharding with chunked broadcasts).

How it works here:

* Every shard's flat parameter buffer is described by *pieces* ``(local_offset, full_offset, length)``:
  a contiguous run of the local buffer that is a contiguous run of the full (unsharded) parameter.
  Row-partitioned weights (fused qkv, gate|up) contribute one piece per section, column-partitioned
  weights (o, down) one piece per row, vocabulary shards and replicated weights a single piece. The layout of ANY shard can be
  computed by any process from the model config alone (the module is built on the meta device).
* Intersecting destination pieces with the (de-duplicated) source pieces of a parameter gives transfer
  segments. Segments of one (source shard, destination shard) pair are packed with the interval-gather
  kernel, moved with ONE point-to-point message and unpacked with the interval-scatter kernel
  (``ops.slice_intervals`` / ``ops.set_intervals``, the HIP counterparts of ``csrc/interval_op``).
* All pairs are posted as one batch of sends / receives: on an xGMI node every pair has its own link, so
  destination replica d pulls from source replica ``d % dp_src`` and all links carry traffic at once
  (the reference funnels through per-pair broadcast groups, one sender per node).
"""
import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from ..models.config import ModelConfig
from ..models.transformer import Transformer
from .pipeline import stage_layer_range


@dataclasses.dataclass(frozen=True)
class Layout:
    dp: int = 1
    tp: int = 1
    pp: int = 1
    vocab_parallel: bool = True  # embedding / LM head rows split over tp (the trainer's layout under tp > 1)
    vpp: int = 1  # interleaved pipeline: every pp rank hosts `vpp` slices of the layers (virtual stages)

    @property
    def world(self) -> int:
        return self.dp * self.tp * self.pp

    def index_of(self, pp: int, dp: int, tp: int) -> int:
        """Position in the model's rank list (same order as `ProcessGrid` with cp = 1)."""
        return (pp * self.dp + dp) * self.tp + tp


def _meta_shard(cfg: ModelConfig, lay: Layout, tp_rank: int, pp_rank: int, dtype=torch.bfloat16) -> Transformer:
    lr, chunks = stage_layer_range(cfg.n_layers, lay.pp, pp_rank), None
    if lay.vpp > 1:  # virtual stage c of rank p is slice c * pp + p of pp * vpp slices
        lr, chunks = None, [stage_layer_range(cfg.n_layers, lay.pp * lay.vpp, c * lay.pp + pp_rank)
                            for c in range(lay.vpp)]
    m = Transformer(cfg, dtype=dtype, device="meta", tp_size=lay.tp, tp_rank=tp_rank, layer_range=lr,
                    layer_chunks=chunks, has_embed=pp_rank == 0, has_head=pp_rank == lay.pp - 1,
                    vocab_parallel=lay.vocab_parallel)
    m.flatten()
    return m


def shard_pieces(model: Transformer, fp=None) -> Dict[str, np.ndarray]:
    """Global parameter name -> int64 array [n, 3] of (local_offset, full_offset, length) pieces, for the
    parameters of the flat buffer `fp` (the model's dense buffer by default; pass ``model.flat_experts`` for
    the routed experts of an expert-parallel shard)."""
    cfg = model.cfg
    fp = model.flat if fp is None else fp
    tp, tr = model.tp_size, model.tp_rank
    etr = model.etp_rank  # routed experts are split over the expert tensor-parallel ranks (tp, or nobody)
    D, H = cfg.head_dim, cfg.hidden_size
    out: Dict[str, np.ndarray] = {}

    def rows(off, row_lo, n_rows, width):  # a block of whole rows is contiguous in both spaces
        return [(off, row_lo * width, n_rows * width)]

    def cols(off, n_rows, local_w, full_w, col_lo):  # one piece per row
        r = np.arange(n_rows, dtype=np.int64)
        return np.stack([off + r * local_w, r * full_w + col_lo, np.full(n_rows, local_w, dtype=np.int64)], 1)

    for n, (off, numel) in fp.offsets.items():
        parts = n.split(".")
        leaf = parts[-1]
        if parts[0] == "layers":
            li = int(parts[1])
            blk = model.layers[li]
            gname = ".".join(["layers", str(model.layer_ids[li])] + parts[2:])
        else:
            blk, gname = None, n
        if leaf in ("qkv", "qkv_bias"):
            hq, hkv = blk.attn.hq, blk.attn.hkv
            w = H if leaf == "qkv" else 1
            q_lo = tr * hq
            kv_lo = tr * hkv if cfg.n_kv_heads >= tp else tr // (tp // cfg.n_kv_heads)
            Q, KV = cfg.n_q_heads, cfg.n_kv_heads
            p = rows(off, q_lo * D, hq * D, w)
            p += rows(off + hq * D * w, Q * D + kv_lo * D, hkv * D, w)
            p += rows(off + (hq + hkv) * D * w, (Q + KV) * D + kv_lo * D, hkv * D, w)
            out[gname] = np.asarray(p, dtype=np.int64)
        elif leaf == "o":
            lw = blk.attn.hq * D
            out[gname] = cols(off, H, lw, cfg.n_q_heads * D, tr * lw)
        elif leaf == "gate_up" and cfg.n_experts == 0:
            f, F_ = blk.mlp.gate_up.shape[0] // 2, cfg.intermediate_size
            out[gname] = np.asarray(rows(off, tr * f, f, H) + rows(off + f * H, F_ + tr * f, f, H), dtype=np.int64)
        elif leaf == "down" and cfg.n_experts == 0:
            f = blk.mlp.down.shape[1]
            out[gname] = cols(off, H, f, cfg.intermediate_size, tr * f)
        elif leaf in ("fc", "fc_bias"):  # ungated MLP (GPT-2): column-parallel rows, their bias with them
            f, w = blk.mlp.fc.shape[0], (H if leaf == "fc" else 1)
            out[gname] = np.asarray(rows(off, tr * f, f, w), dtype=np.int64)
        elif leaf == "proj":  # its row-parallel output projection
            f = blk.mlp.proj.shape[1]
            out[gname] = cols(off, H, f, cfg.intermediate_size, tr * f)
        elif leaf == "gate_up":  # routed experts [E_local, 2f, H]
            E, two_f, _ = blk.mlp.gate_up.shape
            f, F_ = two_f // 2, cfg.moe_intermediate_size
            e0 = blk.mlp.ep_rank * blk.mlp.n_local
            p = []
            for e in range(E):
                base_l, base_f = off + e * two_f * H, (e0 + e) * 2 * F_ * H
                p += [(base_l, base_f + etr * f * H, f * H), (base_l + f * H, base_f + (F_ + etr * f) * H, f * H)]
            out[gname] = np.asarray(p, dtype=np.int64)
        elif leaf == "down":
            E, _, f = blk.mlp.down.shape
            F_ = cfg.moe_intermediate_size
            e0 = blk.mlp.ep_rank * blk.mlp.n_local
            out[gname] = np.concatenate([cols(off + e * H * f, H, f, F_, etr * f) + np.array([0, (e0 + e) * H * F_, 0])
                                         for e in range(E)])
        elif model.vocab_parallel and (leaf == "embed" or (leaf == "head" and not cfg.is_critic)):
            v_lo, v_hi = model.vocab_range
            out[gname] = np.asarray(rows(off, v_lo, v_hi - v_lo, H), dtype=np.int64)
        else:  # replicated over tp: norms, position embeddings, value head, router (and unsharded vocabularies)
            out[gname] = np.asarray([(off, 0, numel)], dtype=np.int64)
    return out


@dataclasses.dataclass
class Transfer:
    src: int  # index into the source model's rank list
    dst: int  # index into the destination model's rank list
    src_intervals: np.ndarray  # [n, 2] half-open ranges of the source shard's flat buffer
    dst_intervals: np.ndarray  # [n, 2] matching ranges of the destination shard's flat buffer

    @property
    def numel(self) -> int:
        return int((self.src_intervals[:, 1] - self.src_intervals[:, 0]).sum())


def _intersect(src_pieces: np.ndarray, src_owner: np.ndarray, dst_pieces: np.ndarray):
    """Overlaps of destination pieces with a sorted, non-overlapping list of source pieces.
    -> (owner, src_local_off, dst_local_off, length) arrays."""
    s_start, s_end = src_pieces[:, 1], src_pieces[:, 1] + src_pieces[:, 2]
    d_start, d_end = dst_pieces[:, 1], dst_pieces[:, 1] + dst_pieces[:, 2]
    lo = np.searchsorted(s_end, d_start, side="right")
    hi = np.searchsorted(s_start, d_end, side="left")
    cnt = np.maximum(hi - lo, 0)
    di = np.repeat(np.arange(len(dst_pieces)), cnt)
    si = np.concatenate([np.arange(a, b) for a, b in zip(lo, hi)]) if cnt.sum() else np.zeros(0, dtype=np.int64)
    ov_s = np.maximum(s_start[si], d_start[di])
    ov_e = np.minimum(s_end[si], d_end[di])
    keep = ov_e > ov_s
    si, di, ov_s, ov_e = si[keep], di[keep], ov_s[keep], ov_e[keep]
    covered = np.zeros(len(dst_pieces), dtype=np.int64)
    np.add.at(covered, di, ov_e - ov_s)
    if not np.array_equal(covered, dst_pieces[:, 2]):
        raise RuntimeError("source shards do not cover a destination parameter slice")
    return (src_owner[si], src_pieces[si, 0] + (ov_s - s_start[si]), dst_pieces[di, 0] + (ov_s - d_start[di]),
            ov_e - ov_s)


def _match(dpieces: np.ndarray, cands: Sequence[np.ndarray]) -> Dict[int, np.ndarray]:
    """Cover the destination pieces of one parameter from the pieces candidate source shards hold of it (in
    order of preference: a range several candidates hold is taken from the first).
    -> candidate index -> int64 [n, 3] of (source local offset, destination local offset, length)."""
    seen, rows_, owner = set(), [], []
    for ci, pieces in enumerate(cands):
        for pc in pieces:
            key = (int(pc[1]), int(pc[2]))
            if key in seen:
                continue
            seen.add(key)
            rows_.append(pc)
            owner.append(ci)
    sp = np.asarray(rows_, dtype=np.int64)
    ow = np.asarray(owner, dtype=np.int64)
    order = np.argsort(sp[:, 1], kind="stable")
    sp, ow = sp[order], ow[order]
    o, so, do, ln = _intersect(sp, ow, dpieces[np.argsort(dpieces[:, 1], kind="stable")])
    return {int(ci): np.stack([so[o == ci], do[o == ci], ln[o == ci]], 1) for ci in np.unique(o)}


def derive_realloc_plan(cfg: ModelConfig, src: Layout, dst: Layout, dtype=torch.bfloat16) -> List[Transfer]:
    """All transfers that turn the source layout's weights into the destination layout's, in a fixed order
    every process agrees on. Destination replica d pulls from source replica d % dp_src."""
    src_shards = {(p, t): shard_pieces(_meta_shard(cfg, src, t, p, dtype)) for p in range(src.pp) for t in range(src.tp)}
    tied = cfg.tie_word_embeddings and not cfg.is_critic
    plan: List[Transfer] = []
    for dpp in range(dst.pp):
        for dtp in range(dst.tp):
            need = shard_pieces(_meta_shard(cfg, dst, dtp, dpp, dtype))
            # per source shard (pp, tp): segments for this destination shard
            seg: Dict[Tuple[int, int], List[np.ndarray]] = {}
            for gname, dpieces in need.items():
                cands = [(p, t) for (p, t), d in src_shards.items() if gname in d]
                lookup = gname
                if not cands and gname == "head" and tied:  # tied LM head: the embedding is the source
                    lookup = "embed"
                    cands = [(p, t) for (p, t), d in src_shards.items() if lookup in d]
                if not cands:
                    raise RuntimeError(f"parameter {gname} does not exist in the source layout")
                # replicated pieces: prefer the source tp rank aligned with this destination rank
                start = (dtp * src.tp) // dst.tp
                cands.sort(key=lambda pt: (pt[0], (pt[1] - start) % src.tp))
                for ci, a in _match(dpieces, [src_shards[pt][lookup] for pt in cands]).items():
                    seg.setdefault(cands[ci], []).append(a)
            for (p, t), chunks in sorted(seg.items()):
                a = np.concatenate(chunks)
                a = a[np.argsort(a[:, 1], kind="stable")]
                for ddp in range(dst.dp):
                    sdp = ddp % src.dp
                    plan.append(Transfer(src=src.index_of(p, sdp, t), dst=dst.index_of(dpp, ddp, dtp),
                                         src_intervals=np.stack([a[:, 0], a[:, 0] + a[:, 2]], 1),
                                         dst_intervals=np.stack([a[:, 1], a[:, 1] + a[:, 2]], 1)))
    plan.sort(key=lambda x: (x.dst, x.src))
    return plan


def _layout_sources(cfg: ModelConfig, src: Dict[str, int], dtype) -> List[Tuple[int, int, Optional[int], Dict[str, np.ndarray]]]:
    """Every distinct parameter buffer of a trainer laid out as `src` (tp, pp, vpp, ep, etp):
    (pipeline stage, tp rank, expert-parallel rank or None for the dense buffer, pieces by parameter name).
    Without expert parallelism the routed experts are part of the dense buffers; with it every (stage, expert
    tensor-parallel rank, expert-parallel rank) has a buffer of its own."""
    tp, pp, vpp = int(src["tp"]), int(src["pp"]), int(src.get("vpp", 1) or 1)
    ep = int(src.get("ep", 1) or 1) if cfg.n_experts else 1
    etp = int(src.get("etp", tp) or tp) if cfg.n_experts else tp
    vocab_parallel = bool(src.get("vocab_parallel", tp > 1))

    def twin(p, t, e):
        lr, chunks = stage_layer_range(cfg.n_layers, pp, p), None
        if vpp > 1:
            lr, chunks = None, [stage_layer_range(cfg.n_layers, pp * vpp, c * pp + p) for c in range(vpp)]
        m = Transformer(cfg, dtype=dtype, device="meta", tp_size=tp, tp_rank=t, layer_range=lr, layer_chunks=chunks,
                        has_embed=p == 0, has_head=p == pp - 1, vocab_parallel=vocab_parallel, ep_size=ep, ep_rank=e,
                        etp_size=etp)
        m.flatten(expert_pad_to=1 if ep > 1 else None)
        return m

    out = []
    for p in range(pp):
        for t in range(tp):
            m = twin(p, t, 0)
            out.append((p, t, None, shard_pieces(m)))
            if ep > 1 and (etp == tp or t == 0):
                for e in range(ep):
                    me = m if e == 0 else twin(p, t, e)
                    out.append((p, t, e, shard_pieces(me, me.flat_experts)))
    return out


def _cover_from_sources(cfg: ModelConfig, sources, src_tp: int, model: Transformer) -> List[Tuple[int, str, np.ndarray]]:
    """Cover the flat buffers of `model` from `sources` (``_layout_sources``).
    -> [(index into sources, destination buffer "flat" | "flat_experts", int64 [n, 3] of (offset in the source
    buffer, offset in the destination buffer, length))], sorted by source."""
    tied = cfg.tie_word_embeddings and not cfg.is_critic
    seg: Dict[Tuple[int, str], List[np.ndarray]] = {}
    for which, fp in (("flat", model.flat), ("flat_experts", model.flat_experts)):
        if fp is None:
            continue
        for gname, dpieces in shard_pieces(model, fp).items():
            lookup = gname
            cands = [i for i, sc in enumerate(sources) if lookup in sc[3]]
            if not cands and gname == "head" and tied:  # tied LM head: the embedding is the source
                lookup = "embed"
                cands = [i for i, sc in enumerate(sources) if lookup in sc[3]]
            if not cands:
                raise RuntimeError(f"parameter {gname} does not exist in the source layout")
            start = (model.tp_rank * src_tp) // model.tp_size  # replicated ranges: spread the reads over the sources
            cands.sort(key=lambda i: (sources[i][0], (sources[i][1] - start) % src_tp, i))
            for ci, a in _match(dpieces, [sources[i][3][lookup] for i in cands]).items():
                seg.setdefault((cands[ci], which), []).append(a)
    out = []
    for (i, which), chunks in sorted(seg.items()):
        a = np.concatenate(chunks)
        out.append((i, which, a[np.argsort(a[:, 1], kind="stable")]))
    return out


def derive_file_load_plan(cfg: ModelConfig, src: Dict[str, int], model: Transformer
                          ) -> List[Tuple[str, str, str, np.ndarray]]:
    """What ONE rank reads from the files of a sharded checkpoint that was written under another layout.

    `src`: the ``layout`` entry of the checkpoint's manifest (tp, pp, vpp, ep, etp); `model`: this rank's
    (flattened) shard. -> [(file name, tensor key in that file, destination buffer ``"flat"`` |
    ``"flat_experts"``, int64 [n, 3] of (offset in the file's tensor, offset in the destination buffer,
    length), in elements)]. Routed experts are covered in every combination: a checkpoint written with expert
    parallelism keeps them in their own files (one per pipeline stage, expert tensor-parallel rank and
    expert-parallel rank), one written without keeps them inside the dense shards; the reader may hold them
    either way (resharding load of ``areal/experimental/utils/megatron_checkpointer.py``)."""
    sources = _layout_sources(cfg, dict(src, vocab_parallel=int(src["tp"]) > 1), model.dtype)
    out = []
    for i, which, tri in _cover_from_sources(cfg, sources, int(src["tp"]), model):
        p, t, e, _ = sources[i]
        if e is None:
            out.append((f"shard-pp{p}-tp{t}.safetensors", "flat", which, tri))
        else:
            out.append((f"experts-pp{p}-tp{t}-ep{e}.safetensors", "flat_experts", which, tri))
    return out


def derive_expert_parallel_pull_plan(cfg: ModelConfig, src: Dict[str, int], dst_tp: int, dst_tp_rank: int,
                                     replica: int = 0, dtype=torch.bfloat16) -> List[Tuple[int, str, np.ndarray]]:
    """Pull plan of a rollout-server shard against a trainer WITH expert parallelism: its ranks export two buffers
    each, the dense parameters (the same over the data-parallel ranks of a (pp, tp) slice) and the routed experts
    they hold (which differ over the data-parallel ranks: expert-parallel rank = position in the expert grid of
    the stage modulo ep). `src`: dp, cp, tp, pp, vpp, ep, etp, vocab_parallel of the trainer.
    -> [(trainer rank in grid order, "flat" | "flat_experts" (which of that rank's buffers), int64 [n, 3] of
    (offset in that buffer, offset in the server shard's flat buffer, length))]. Servers spread over the replicas
    that hold the same shard by `replica`."""
    tp, pp = int(src["tp"]), int(src["pp"])
    cp = int(src.get("cp", 1) or 1)
    dp = int(src["dp"]) // cp  # `dp` counts the ranks that hold the same dense slice: data x context parallel
    ep, etp = int(src["ep"]), int(src.get("etp", tp) or tp)
    key = ("ep", repr(cfg), repr(sorted(src.items())), dst_tp, dst_tp_rank, replica, str(dtype))
    if key in _PULL_PLANS:
        return _PULL_PLANS[key]
    sources = _layout_sources(cfg, src, dtype)
    dst = _meta_shard(cfg, Layout(dp=1, tp=dst_tp, pp=1, vocab_parallel=False), dst_tp_rank, 0, dtype)

    def rank_of(p, d, t):
        return ((p * dp + d) * cp) * tp + t

    out = []
    for i, _which, tri in _cover_from_sources(cfg, sources, tp, dst):
        p, t, e, _ = sources[i]
        if e is None:
            r, buf = rank_of(p, replica % dp, t), "flat"
        elif etp == tp:  # expert grid of a stage = its data-parallel ranks
            r, buf = rank_of(p, e + ep * (replica % max(1, dp // ep)), t), "flat_experts"
        else:  # whole experts: the grid runs over (dp x tp), tp fastest
            slots = [s for s in range(dp * tp) if s % ep == e]
            d, tt = divmod(slots[replica % len(slots)], tp)
            r, buf = rank_of(p, d, tt), "flat_experts"
        siv, div = _merged(np.stack([tri[:, 0], tri[:, 0] + tri[:, 2]], 1), np.stack([tri[:, 1], tri[:, 1] + tri[:, 2]], 1))
        out.append((r, buf, np.stack([siv[:, 0], div[:, 0], siv[:, 1] - siv[:, 0]], 1).astype(np.int64)))
    _PULL_PLANS[key] = out
    return out


def _merged(iv_src: np.ndarray, iv_dst: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Fuse neighbouring segments that are contiguous on both sides (fewer, longer copies)."""
    if len(iv_src) <= 1:
        return iv_src, iv_dst
    brk = (iv_src[1:, 0] != iv_src[:-1, 1]) | (iv_dst[1:, 0] != iv_dst[:-1, 1])
    starts = np.concatenate([[0], np.nonzero(brk)[0] + 1])
    ends = np.concatenate([starts[1:], [len(iv_src)]]) - 1
    return (np.stack([iv_src[starts, 0], iv_src[ends, 1]], 1), np.stack([iv_dst[starts, 0], iv_dst[ends, 1]], 1))


@torch.no_grad()
def realloc_parameters(plan: Sequence[Transfer], src_ranks: Sequence[int], dst_ranks: Sequence[int],
                       src_flat: Optional[torch.Tensor], dst_flat: Optional[torch.Tensor], group=None,
                       eta: float = 1.0) -> int:
    """Execute this process's part of `plan`. `src_ranks` / `dst_ranks`: global ranks of the two models in
    layout order; `src_flat` / `dst_flat`: this process's flat buffers of the source / destination model
    (None when it hosts no shard of that model). ``eta < 1`` mixes instead of overwriting:
    ``dst = eta * src + (1 - eta) * dst`` (exponential moving average targets). Returns elements received."""
    me = dist.get_rank() if dist.is_initialized() else 0
    p2p, recvs, keep = [], [], []
    n_recv = 0
    for tag, t in enumerate(plan):
        s_rank, d_rank = src_ranks[t.src], dst_ranks[t.dst]
        if me != s_rank and me != d_rank:
            continue
        siv, div = _merged(t.src_intervals, t.dst_intervals)
        if me == s_rank and me == d_rank:
            packed = ops.slice_intervals(src_flat, siv)
            recvs.append((packed, div))
        elif me == s_rank:
            packed = ops.slice_intervals(src_flat, siv)
            keep.append(packed)
            p2p.append(dist.P2POp(dist.isend, packed, d_rank, group=group, tag=tag))
        else:
            buf = torch.empty(t.numel, dtype=dst_flat.dtype, device=dst_flat.device)
            p2p.append(dist.P2POp(dist.irecv, buf, s_rank, group=group, tag=tag))
            recvs.append((buf, div))
    if p2p:
        for w in dist.batch_isend_irecv(p2p):
            w.wait()
    for buf, div in recvs:
        buf = buf.to(dst_flat.dtype)
        if eta != 1.0:
            cur = ops.slice_intervals(dst_flat, div)
            buf = (eta * buf.float() + (1.0 - eta) * cur.float()).to(dst_flat.dtype)
        ops.set_intervals(buf, dst_flat, div)
        n_recv += buf.numel()
    return n_recv


# ------------------------------------------------------------------------------------------------------------------
# Pull plans: what ONE shard of a rollout server copies out of the exported buffers of the trainer's ranks when the
# two sides are laid out differently (peer-memory weight updates, ``ops/ipc.py``). Same piece algebra as above, but
# nobody sends: the destination reads the source ranks' memory directly.
# ------------------------------------------------------------------------------------------------------------------
_PULL_PLANS: Dict[Tuple, List[Tuple[int, np.ndarray]]] = {}


def derive_pull_plan(cfg: ModelConfig, src: Layout, dst_tp: int, dst_tp_rank: int, replica: int = 0,
                     dtype=torch.bfloat16, zero3: int = 1) -> List[Tuple[int, np.ndarray]]:
    """-> [(trainer rank in layout order, int64 [n, 3] (offset in that rank's exported buffer, offset in the
    destination shard's flat buffer, length), in elements)] for tensor-parallel rank `dst_tp_rank` of a rollout
    server with `dst_tp` ranks (unsharded vocabulary, no pipeline stages). The server pulls from data-parallel
    replica ``replica % src.dp`` of the trainer, so that several servers spread over the trainer's replicas (and
    their xGMI links). ``zero3 = n``: every (pp, tp) shard of the trainer is itself cut over the n ranks of its
    data-parallel group (ZeRO-3, ``parallel/zero3.py``): all of them are sources, each for the part it holds.
    Every trainer GPU thus serves exactly the bytes it owns: the role of the sharded push that replaces the
    rank-0 funnel of ``areal/engine/fsdp_engine.py:399-433``."""
    key = (repr(cfg), src, dst_tp, dst_tp_rank, replica % max(1, src.dp), str(dtype), zero3)
    if key in _PULL_PLANS:
        return _PULL_PLANS[key]
    dst = Layout(dp=1, tp=dst_tp, pp=1, vocab_parallel=False)
    src1 = dataclasses.replace(src, dp=1)
    out: List[Tuple[int, np.ndarray]] = []
    for t in derive_realloc_plan(cfg, src1, dst, dtype):
        if t.dst != dst_tp_rank:
            continue
        p, tr = divmod(t.src, src.tp)
        if zero3 <= 1:
            siv, div = _merged(t.src_intervals, t.dst_intervals)
            out.append((src.index_of(p, replica % src.dp, tr),
                        np.stack([siv[:, 0], div[:, 0], siv[:, 1] - siv[:, 0]], 1).astype(np.int64)))
            continue
        assert zero3 == src.dp, "ZeRO-3 shards a (pp, tp) slice over ALL data-parallel ranks"
        from .zero3 import shard_map

        per_rank = shard_map(_meta_shard(cfg, src1, tr, p, dtype), zero3).translate(t.src_intervals, t.dst_intervals)
        for z, tri in per_rank.items():
            out.append((src.index_of(p, z, tr), tri))
    _PULL_PLANS[key] = out
    return out
