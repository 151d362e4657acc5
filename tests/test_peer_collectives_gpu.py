"""Collectives over peer-mapped memory (`ops/peer_collectives.py`) between processes that share this GPU, against the
same exchanges computed on the host from every rank's input."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(world, shape, call, dtype):
    g = torch.Generator().manual_seed(7919 * call + 13)
    return [torch.randn(*shape, generator=g).to(dtype) for _ in range(world)]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    from areal_vllm_amd.ops.peer_collectives import PeerCollectives, open_for

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # a slot of 192 KiB: the larger messages below need several kernels each
    pc = open_for(dist.group.WORLD, dev, 192 << 10)
    assert isinstance(pc, PeerCollectives) and pc.slot_bytes == 192 << 10
    n, call, kept = world, 0, []
    for rep in range(3):
        # Ulysses exchange: [t, n, h, D] <-> [n t, h, D]
        for t, h, D in [(5, 1, 8), (37, 3, 128), (300, 2, 128)]:
            xs = _inputs(n, (t, n, h, D), call, torch.bfloat16)
            y = pc.seq_to_head(xs[rank].to(dev))
            want = torch.cat([x[:, rank] for x in xs], 0)
            assert torch.equal(y.cpu(), want), ("seq_to_head", rank, t, h, D)
            back = pc.head_to_seq(y)
            assert torch.equal(back.cpu(), xs[rank]), ("head_to_seq", rank, t, h, D)
            kept.append(y.cpu())
            call += 1
        # all-gather / reduce-scatter of flat buffers, fp32 and bf16, also in place on the rank's own slice
        for numel, dtype in [(8, torch.float32), (1000, torch.float32), (65536 + 24, torch.float32),
                             (4096, torch.bfloat16), (200000, torch.bfloat16)]:
            xs = _inputs(n, (n * numel,), call, dtype)
            out = torch.empty(numel, dtype=dtype, device=dev)
            pc.reduce_scatter_into(out, xs[rank].to(dev))
            acc = torch.zeros(numel)
            for x in xs:  # rank order, fp32 accumulation
                acc += x[rank * numel:(rank + 1) * numel].float()
            assert torch.equal(out.cpu(), acc.to(dtype)), ("reduce_scatter", rank, numel, dtype)
            full = torch.zeros(n * numel, dtype=dtype, device=dev)
            full[rank * numel:(rank + 1) * numel] = out
            pc.all_gather_into(full, full[rank * numel:(rank + 1) * numel])
            want = torch.zeros(n * numel)
            for x in xs:
                want += x.float()
            assert torch.equal(full.cpu(), want.to(dtype)), ("all_gather", rank, numel, dtype)
            kept.append(full.cpu())
            call += 1
        # rows (sequence parallelism)
        xs = _inputs(n, (n * 33, 96), call, torch.bfloat16)
        got = pc.reduce_scatter_rows(xs[rank].to(dev))
        want = sum(x[rank * 33:(rank + 1) * 33].float() for x in xs).to(torch.bfloat16)
        assert torch.equal(got.cpu(), want)
        gathered = pc.all_gather_rows(got)
        assert torch.equal(gathered.cpu(), sum(x.float() for x in xs).to(torch.bfloat16))
        call += 1
    # a bounded launch (beside compute kernels) gives the same result
    xs = _inputs(n, (n * 300000,), call, torch.float32)
    out = torch.empty(300000, device=dev)
    pc.reduce_scatter_into(out, xs[rank].to(dev), max_blocks=4)
    acc = torch.zeros(300000)
    for x in xs:
        acc += x[rank * 300000:(rank + 1) * 300000]
    assert torch.equal(out.cpu(), acc)
    # graph replay: the sequence number lives in device memory
    x = torch.zeros(16, n, 2, 64, dtype=torch.bfloat16, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pc.seq_to_head(x)  # warm-up launch outside the capture, on every rank alike
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    y = torch.empty(n * 16, 2, 64, dtype=torch.bfloat16, device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y.copy_(pc.seq_to_head(x))
    for rep in range(3):
        x.fill_(float(rank + 1 + rep))
        graph.replay()
        torch.cuda.synchronize()
        want = torch.arange(1, n + 1).float().add(rep).repeat_interleave(16)
        assert torch.equal(y[:, 0, 0].float().cpu(), want), (rank, rep)
    assert not pc.error()
    # sequence-parallel linears pipelined with their collective (an exchange object of their own, side stream)
    from areal_vllm_amd.parallel.sp_overlap import SPOverlap, gather_linear, linear_scatter

    pc2 = open_for(dist.group.WORLD, dev, 192 << 10)
    ov = SPOverlap(pc2, chunks=3, min_rows=32)
    before = pc2.calls  # the start-up self-check
    t, H, Fc = 300, 64, 96
    assert len(ov.plan(t, H * 2)) == 3
    xs = _inputs(n, (t, H), 901, torch.bfloat16)
    w1s = [w * 0.2 for w in _inputs(n, (Fc, H), 902, torch.bfloat16)]
    w2s = [w * 0.2 for w in _inputs(n, (H, Fc), 903, torch.bfloat16)]
    gs = _inputs(n, (t, H), 904, torch.float32)
    x = xs[rank].to(dev).requires_grad_(True)
    w1 = w1s[rank].to(dev).requires_grad_(True)
    w2 = w2s[rank].to(dev).requires_grad_(True)
    z = linear_scatter(gather_linear(x, w1, ov), w2, ov)
    (z.float() * gs[rank].to(dev)).sum().backward()
    torch.cuda.synchronize()
    X = torch.cat(xs).float().requires_grad_(True)
    W1 = [w.float().requires_grad_(True) for w in w1s]
    W2 = [w.float().requires_grad_(True) for w in w2s]
    Z = sum((X @ a.t()).bfloat16().float() @ b.t() for a, b in zip(W1, W2))
    (Z * torch.cat(gs)).sum().backward()
    rows = slice(rank * t, (rank + 1) * t)

    def close(got, want, what):
        err = (got.float().cpu() - want).abs().max().item()
        assert err <= 3e-2 * want.abs().max().item() + 1e-3, (what, rank, err, want.abs().max().item())

    close(z, Z.detach()[rows], "forward")
    close(x.grad, X.grad[rows], "dx")
    close(w1.grad, W1[rank].grad, "dw1")
    close(w2.grad, W2[rank].grad, "dw2")
    assert not pc2.error()
    assert pc2.calls - before == 4 * 3 + 3, pc2.calls  # four pipelined collectives + the re-gather for dW1
    torch.save(kept, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    pc.close()
    pc2.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_peer_memory_exchange_kernels_between_processes(tmp_path, world):
    """Ulysses sequence<->head exchange (both directions), all-gather and rank-ordered reduce-scatter (fp32, bf16; flat
    and by rows), chunked through a small staging slot, with a bounded grid, and replayed from a captured graph."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    # all-gathered buffers carry the same bits on every rank
    first, *others = (torch.load(tmp_path / f"r{r}.pt", weights_only=True) for r in range(world))
    for other in others:
        assert all(torch.equal(x, y) for x, y in zip(first[3:8], other[3:8]))
