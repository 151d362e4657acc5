"""Pipeline parallelism: layer partition and the 1F1B micro-batch schedule.

Stages exchange one stacked ``[2, T, H]`` tensor (block output + residual stream) per micro-batch
through batched point-to-point ops, so neighbouring stages talk over a single xGMI link with one
message per direction per step. The schedule is the classic one-forward-one-backward order
(warm-up forwards = stages ahead of me, steady 1F1B, cool-down backwards) that the reference gets from
its instruction-based pipeline runner (``realhf/impl/model/backend/pipe_runner.py:29-1045``,
``.../static_schedule.py:1-417``) and from Megatron's ``forward_backward_func``
(``areal/experimental/megatron_engine.py:476-673``); here it is a plain loop over micro-batches - there
is no instruction interpreter because every stage is the same SPMD program.

Every pipeline stage of one data-parallel replica receives the same input batch, therefore all stages
derive identical micro-batch lists (shapes of the exchanged tensors are known without a handshake).
"""
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import comm


def stage_layer_range(n_layers: int, pp: int, stage: int) -> Tuple[int, int]:
    """Contiguous, balanced layer slice of `stage`: the first `n_layers % pp` stages take one extra layer."""
    base, rem = divmod(n_layers, pp)
    lo = stage * base + min(stage, rem)
    hi = lo + base + (1 if stage < rem else 0)
    return lo, hi


# ------------------------------------------------------------------------------ p2p helpers
def _exchange(grid, send_next=None, send_prev=None, recv_prev_like=None, recv_next_like=None):
    """One batched round of sends / receives with the neighbouring stages; returns (from_prev, from_next)."""
    ops = []
    from_prev = from_next = None
    if recv_prev_like is not None:
        from_prev = torch.empty(recv_prev_like[0], dtype=recv_prev_like[1], device=recv_prev_like[2])
        ops.append(dist.P2POp(dist.irecv, from_prev, grid.prev_stage_rank()))
    if send_next is not None:
        ops.append(dist.P2POp(dist.isend, send_next.contiguous(), grid.next_stage_rank()))
    if send_prev is not None:
        ops.append(dist.P2POp(dist.isend, send_prev.contiguous(), grid.prev_stage_rank()))
    if recv_next_like is not None:
        from_next = torch.empty(recv_next_like[0], dtype=recv_next_like[1], device=recv_next_like[2])
        ops.append(dist.P2POp(dist.irecv, from_next, grid.next_stage_rank()))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return from_prev, from_next


def _act_like(engine, pmb) -> Tuple[Tuple[int, int, int], torch.dtype, torch.device]:
    T = int(pmb["input_ids"].shape[0])
    if engine.model.sp is not None:
        T //= engine.model.sp.size
    if engine.model.seq_par:  # token-sharded residual stream between tensor-parallel regions
        T //= engine.model.tp_size
    return (2, T, engine.model_config.hidden_size), engine.model.dtype, engine.device


def sync_tied_embedding_grads(engine):
    """Tied input/output embeddings live on the first and last stage: sum their gradients so both copies
    take the same optimizer step (reference: ``realhf/impl/model/backend/pipe_runner.py`` tied-weight
    all-reduce; Megatron's embedding group)."""
    g = engine.grid
    if g.pp == 1 or not engine.model_config.tie_word_embeddings or engine.model_config.is_critic:
        return
    grp = g.group("embed")
    if grp is None:
        return
    m = engine.model
    p = m.embed if g.is_first_stage() else m.head
    dist.all_reduce(p.main_grad, group=grp)


def sync_tied_embedding_weights(engine):
    """Copy the first stage's embedding into the last stage's LM head (after random initialisation)."""
    g = engine.grid
    if g.pp == 1 or not engine.model_config.tie_word_embeddings or engine.model_config.is_critic:
        return
    grp = g.group("embed")
    if grp is None:
        return
    m = engine.model
    p = m.embed if g.is_first_stage() else m.head
    dist.broadcast(p.data, src=g.group_ranks["embed"][0], group=grp)


# ------------------------------------------------------------------------------- schedules
def pipeline_train_batch(engine, input_: Dict[str, Any], loss_fn: Callable, loss_weight_fn: Callable):
    g = engine.grid
    if g.vpp > 1:
        return interleaved_train_batch(engine, input_, loss_fn, loss_weight_fn)
    first, last = g.is_first_stage(), g.is_last_stage()
    dev = engine.device
    engine.zero_main_grads()
    mb_list = engine.prepare_mb_list(input_)
    M = len(mb_list.padded_mbs)
    weights = [loss_weight_fn(mb) for mb in mb_list.mbs]
    total_w = torch.stack([torch.as_tensor(w, dtype=torch.float32, device=dev) for w in weights]).sum()
    comm.all_reduce_sum(total_w, g.group("dp"))
    assert float(total_w) > 0, "total loss weight must be positive"

    inputs: List[Optional[torch.Tensor]] = []
    outputs: List[Any] = []

    def fwd(i: int, buf: Optional[torch.Tensor]):
        pmb = mb_list.padded_mbs[i]
        h_in = r_in = None
        if buf is not None:
            buf.requires_grad_(True)
            h_in, r_in = buf[0], buf[1]
        out = engine._run_model(pmb, h_in, r_in)
        scale = torch.as_tensor(weights[i], dtype=torch.float32, device=dev) / total_w
        aux = engine.model.moe_aux_loss()
        if last:
            proxy = engine._proxy(out, mb_list.padding_lengths[i])
            mb = engine._unpadded(pmb, mb_list.padding_lengths[i], mb_list.old_cu_seqlens_list[i])
            loss = loss_fn(proxy, mb)
            if aux is not None:
                loss = loss + aux
            res = (loss * scale, None)
        else:
            res = (torch.stack([out[0], out[1]]), None if aux is None else aux * scale)
        inputs.append(buf)
        outputs.append(res)
        return res[0]

    def bwd(grad_out: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        buf = inputs.pop(0)
        out, aux = outputs.pop(0)
        if last:
            out.backward()
        elif aux is None:
            torch.autograd.backward([out], [grad_out])
        else:
            torch.autograd.backward([out, aux], [grad_out, None])
        return None if buf is None else buf.grad

    like = [_act_like(engine, pmb) for pmb in mb_list.padded_mbs]
    warm = min(g.pp - g.coord.pp - 1, M)
    steady = M - warm
    nf = 0  # forwards issued
    nb = 0  # backwards issued
    for _ in range(warm):
        buf, _n = _exchange(g, recv_prev_like=None if first else like[nf])
        out = fwd(nf, buf)
        _exchange(g, send_next=None if last else out.detach())
        nf += 1
    buf = None
    if steady > 0 and not first:
        buf, _n = _exchange(g, recv_prev_like=like[nf])
    for k in range(steady):
        out = fwd(nf, buf)
        nf += 1
        grad_out = None
        if not last:
            _p, grad_out = _exchange(g, send_next=out.detach(), recv_next_like=like[nb])
        grad_in = bwd(grad_out)
        nb += 1
        more = k < steady - 1
        if first:
            buf = None
        else:
            buf, _n = _exchange(g, send_prev=grad_in, recv_prev_like=like[nf] if more else None)
    for _ in range(warm):
        _p, grad_out = _exchange(g, recv_next_like=None if last else like[nb])
        grad_in = bwd(grad_out)
        nb += 1
        if not first:
            _exchange(g, send_prev=grad_in)
    assert not inputs and not outputs and nf == M and nb == M

    sync_tied_embedding_grads(engine)
    return engine.optimizer_step()


# ------------------------------------------------------------------- interleaved (virtual) stages
class _Ring:
    """Point-to-point traffic of one rank in an interleaved pipeline: every transfer is an asynchronous send
    or receive on the two-rank communicator of its link and direction (``ProcessGrid._build_pipeline_links``).
    On each link the sender's order of sends equals the receiver's order of receives by construction of the
    schedules below, and no rank ever waits for a send to be matched, so the schedules cannot dead-lock on
    communication; they only wait for data they need."""

    def __init__(self, grid):
        self.g = grid
        self.pending = []

    def recv(self, direction: str, like):
        buf = torch.empty(like[0], dtype=like[1], device=like[2])
        src = self.g.ring_prev if direction == "fwd" else self.g.ring_next
        dist.irecv(buf, src=src, group=self.g.links["recv_" + direction]).wait()
        return buf

    def send(self, direction: str, t: torch.Tensor):
        dst = self.g.ring_next if direction == "fwd" else self.g.ring_prev
        t = t.contiguous()
        self.pending.append((dist.isend(t, dst=dst, group=self.g.links["send_" + direction]), t))

    def drain(self):
        for w, _t in self.pending:
            w.wait()
        self.pending.clear()


def _virtual_order(M: int, P: int, V: int, backward: bool) -> List[Tuple[int, int]]:
    """(micro-batch, chunk) in the depth-first order of interleaved schedules: micro-batches go through a
    chunk in groups of P before the group moves on to the next chunk (backward: chunks in reverse)."""
    order = []
    for g0 in range(0, M, P):
        group = range(g0, min(M, g0 + P))
        for c in (range(V - 1, -1, -1) if backward else range(V)):
            order += [(m, c) for m in group]
    return order


def interleaved_train_batch(engine, input_: Dict[str, Any], loss_fn: Callable, loss_weight_fn: Callable):
    """One optimizer step with V layer chunks per rank (virtual stage c * P + r on rank r, chunk c): the bubble of
    the 1F1B schedule shrinks from (P-1)/M to (P-1)/(M*V) of the step at the price of V times as many, V times
    smaller, stage-to-stage messages - cheap on xGMI. Order of work on every rank: warm-up forwards, then one
    forward / one backward alternately, then the remaining backwards, forwards and backwards each in
    depth-first order (the schedule of Megatron's interleaved ``forward_backward_func`` that the reference
    selects with ``virtual_pipeline_parallel_size``, ``areal/experimental/megatron_engine.py:169-179``). When
    the number of micro-batches is not a multiple of P the alternation is not provably free of circular waits:
    such a step runs all forwards, then all backwards."""
    g = engine.grid
    P, V, r = g.pp, g.vpp, g.coord.pp
    dev = engine.device
    engine.zero_main_grads()
    mb_list = engine.prepare_mb_list(input_)
    M = len(mb_list.padded_mbs)
    weights = [loss_weight_fn(mb) for mb in mb_list.mbs]
    total_w = torch.stack([torch.as_tensor(w, dtype=torch.float32, device=dev) for w in weights]).sum()
    comm.all_reduce_sum(total_w, g.group("dp"))
    assert float(total_w) > 0, "total loss weight must be positive"
    like = [_act_like(engine, pmb) for pmb in mb_list.padded_mbs]
    ring = _Ring(g)
    last_vs = P * V - 1
    saved: Dict[Tuple[int, int], Any] = {}

    def fwd(m: int, c: int):
        vs = c * P + r
        buf = None
        if vs > 0:
            buf = ring.recv("fwd", like[m])
            buf.requires_grad_(True)
        pmb = mb_list.padded_mbs[m]
        out = engine._run_model(pmb, None if buf is None else buf[0], None if buf is None else buf[1], chunk=c)
        scale = torch.as_tensor(weights[m], dtype=torch.float32, device=dev) / total_w
        aux = engine.model.moe_aux_loss()
        if vs == last_vs:
            proxy = engine._proxy(out, mb_list.padding_lengths[m])
            mb = engine._unpadded(pmb, mb_list.padding_lengths[m], mb_list.old_cu_seqlens_list[m])
            loss = loss_fn(proxy, mb)
            if aux is not None:
                loss = loss + aux
            saved[(m, c)] = (buf, loss * scale, None)
        else:
            y = torch.stack([out[0], out[1]])
            ring.send("fwd", y.detach())
            saved[(m, c)] = (buf, y, None if aux is None else aux * scale)

    def bwd(m: int, c: int):
        vs = c * P + r
        buf, out, aux = saved.pop((m, c))
        if vs == last_vs:
            out.backward()
        else:
            grad = ring.recv("bwd", like[m])
            if aux is None:
                torch.autograd.backward([out], [grad])
            else:
                torch.autograd.backward([out, aux], [grad, None])
        if buf is not None:
            ring.send("bwd", buf.grad)

    f_order, b_order = _virtual_order(M, P, V, False), _virtual_order(M, P, V, True)
    total = M * V
    warm = min((P - r - 1) * 2 + (V - 1) * P, total) if M % P == 0 else total
    for k in range(warm):
        fwd(*f_order[k])
    for i in range(total - warm):
        fwd(*f_order[warm + i])
        bwd(*b_order[i])
    for i in range(total - warm, total):
        bwd(*b_order[i])
    ring.drain()
    assert not saved
    sync_tied_embedding_grads(engine)
    return engine.optimizer_step()


def _interleaved_forward_only(engine, mb_list, on_last: Callable):
    """Forward passes of all micro-batches through all virtual stages; `on_last(m, hidden)` runs on the rank
    that hosts the last virtual stage."""
    g = engine.grid
    P, V, r = g.pp, g.vpp, g.coord.pp
    ring = _Ring(g)
    for m, c in _virtual_order(len(mb_list.padded_mbs), P, V, False):
        vs = c * P + r
        pmb = mb_list.padded_mbs[m]
        buf = ring.recv("fwd", _act_like(engine, pmb)) if vs > 0 else None
        out = engine._run_model(pmb, None if buf is None else buf[0], None if buf is None else buf[1], chunk=c)
        if vs == P * V - 1:
            on_last(m, out)
        else:
            ring.send("fwd", torch.stack([out[0], out[1]]))
    ring.drain()


@torch.no_grad()
def pipeline_eval_batch(engine, input_, loss_fn, loss_weight_fn):
    g = engine.grid
    first, last = g.is_first_stage(), g.is_last_stage()
    dev = engine.device
    mb_list = engine.prepare_mb_list(input_)
    weights = [loss_weight_fn(mb) for mb in mb_list.mbs]
    total_w = torch.stack([torch.as_tensor(w, dtype=torch.float32, device=dev) for w in weights]).sum()
    comm.all_reduce_sum(total_w, g.group("dp"))
    total = torch.zeros((), dtype=torch.float32, device=dev)
    if g.vpp > 1:
        def on_last(m, out):
            proxy = engine._proxy(out, mb_list.padding_lengths[m])
            mb = engine._unpadded(mb_list.padded_mbs[m], mb_list.padding_lengths[m], mb_list.old_cu_seqlens_list[m])
            total.add_(loss_fn(proxy, mb).float() * torch.as_tensor(weights[m], dtype=torch.float32, device=dev) / total_w)

        _interleaved_forward_only(engine, mb_list, on_last)
    for i, pmb in enumerate(mb_list.padded_mbs if g.vpp == 1 else []):
        buf, _n = _exchange(g, recv_prev_like=None if first else _act_like(engine, pmb))
        out = engine._run_model(pmb, None if buf is None else buf[0], None if buf is None else buf[1])
        if last:
            proxy = engine._proxy(out, mb_list.padding_lengths[i])
            mb = engine._unpadded(pmb, mb_list.padding_lengths[i], mb_list.old_cu_seqlens_list[i])
            total += loss_fn(proxy, mb).float() * torch.as_tensor(weights[i], dtype=torch.float32, device=dev) / total_w
        else:
            _exchange(g, send_next=torch.stack([out[0], out[1]]))
    comm.all_reduce_sum(total, g.group("dp"))
    # every stage returns the value computed on the last one
    dist.broadcast(total, src=g.group_ranks["pp"][-1], group=g.group("pp"))
    return total


@torch.no_grad()
def pipeline_forward(engine, input_, output_seqlens=None, post_hook=None, aggregate_fn=torch.cat):
    from ..utils.data import unpack_sequence

    g = engine.grid
    first, last = g.is_first_stage(), g.is_last_stage()
    B, S = input_["attention_mask"].shape
    mb_list = engine.prepare_mb_list(input_)
    if output_seqlens is None:
        output_seqlens = input_["attention_mask"].sum(-1).long().tolist()
    results = []
    if g.vpp > 1:
        by_mb: Dict[int, Any] = {}

        def on_last(m, out):
            proxy = engine._proxy(out, mb_list.padding_lengths[m])
            mb = engine._unpadded(mb_list.padded_mbs[m], mb_list.padding_lengths[m], mb_list.old_cu_seqlens_list[m])
            by_mb[m] = post_hook(proxy, mb) if post_hook is not None else proxy.materialize()

        _interleaved_forward_only(engine, mb_list, on_last)
        results = [by_mb[m] for m in sorted(by_mb)]
    for i, pmb in enumerate(mb_list.padded_mbs if g.vpp == 1 else []):
        buf, _n = _exchange(g, recv_prev_like=None if first else _act_like(engine, pmb))
        out = engine._run_model(pmb, None if buf is None else buf[0], None if buf is None else buf[1])
        if last:
            proxy = engine._proxy(out, mb_list.padding_lengths[i])
            mb = engine._unpadded(pmb, mb_list.padding_lengths[i], mb_list.old_cu_seqlens_list[i])
            results.append(post_hook(proxy, mb) if post_hook is not None else proxy.materialize())
        else:
            _exchange(g, send_next=torch.stack([out[0], out[1]]))
    # the last stage owns the result; ship its shape, then the data, to the other stages
    src = g.group_ranks["pp"][-1]
    if last:
        res = aggregate_fn(results).contiguous()
        meta = [tuple(res.shape), str(res.dtype).replace("torch.", "")]
    else:
        res, meta = None, None
    box = [meta]
    dist.broadcast_object_list(box, src=src, group=g.group("pp"))
    if not last:
        res = torch.empty(box[0][0], dtype=getattr(torch, box[0][1]), device=engine.device)
    dist.broadcast(res, src=src, group=g.group("pp"))
    lens_fwd = [output_seqlens[i] for i in mb_list.forward_indices]
    seqs = unpack_sequence(res, lens=lens_fwd)
    ordered = [seqs[i] for i in mb_list.backward_indices]
    out = res.new_zeros(B, S, *res.shape[1:])
    for b, s in enumerate(ordered):
        out[b, : s.shape[0]] = s
    return out
