"""Vocabulary-parallel embedding and LM-head loss (two gloo ranks on the CPU) against the unsharded ops.

The sharded path must reproduce the full-vocabulary log-probs, entropies and ALL gradients (hidden states,
head rows, embedding rows), also when the shards differ in size and when labels fall into the other shard.
Reference behaviour: ``realhf/impl/model/parallelism/tensor_parallel/modules.py:57-139, 1060-1177``.
"""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _reference(h, w, emb_ids, labels, temperature, dlogp, dent):
    """Full-vocabulary fp32 reference: tied table `w` used as embedding and as LM head."""
    w = w.clone().requires_grad_(True)
    h = h.clone().requires_grad_(True)
    x = h + torch.nn.functional.embedding(emb_ids, w)
    z = (x @ w.t()) / temperature
    lsm = torch.log_softmax(z, -1)
    logp = lsm.gather(-1, labels.unsqueeze(-1)).squeeze(-1)
    ent = -(lsm.exp() * lsm).sum(-1)
    ((logp * dlogp).sum() + (ent * dent).sum()).backward()
    return logp.detach(), ent.detach(), h.grad, w.grad


def _rank_main(rank, world, port, vocab, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      GLOO_SOCKET_IFNAME="lo")
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    import torch.distributed as dist

    from areal_vllm_amd import ops
    from areal_vllm_amd.parallel import tensor_parallel as tpar

    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    T, H, temp = 37, 16, 0.7
    h = torch.randn(T, H, generator=g)
    w = torch.randn(vocab, H, generator=g) * 0.3
    ids = torch.randint(0, vocab, (T,), generator=g)
    labels = torch.randint(0, vocab, (T,), generator=g)
    labels[0], labels[1] = 0, vocab - 1  # both ends of the vocabulary
    dlogp, dent = torch.randn(T, generator=g), torch.randn(T, generator=g)
    want = _reference(h, w, ids, labels, temp, dlogp, dent)

    lo, hi = ops.vocab_shard_range(vocab, world, rank)
    ws = w[lo:hi].clone().requires_grad_(True)
    hh = h.clone().requires_grad_(True)
    x = hh + tpar.vocab_parallel_embedding(ids, ws, dist.group.WORLD, vocab_start=lo)
    logp, ent = ops.lm_head_logprobs(x, ws, labels, temp, chunk=16, keep_logits=bool(rank % 2), vocab_start=lo,
                                     group=dist.group.WORLD, reduce_dh=True)
    ((logp * dlogp).sum() + (ent * dent).sum()).backward()
    torch.testing.assert_close(logp.detach(), want[0], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(ent.detach(), want[1], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(hh.grad, want[2], atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(ws.grad, want[3][lo:hi], atol=1e-5, rtol=1e-4)

    # full logits from the shards (uneven widths), and their backward
    ws2 = w[lo:hi].clone().requires_grad_(True)
    full = tpar.GatherVocab.apply(torch.nn.functional.linear(h, ws2), dist.group.WORLD, vocab)
    torch.testing.assert_close(full.detach(), h @ w.t(), atol=1e-5, rtol=1e-5)
    full.square().sum().backward()
    torch.testing.assert_close(ws2.grad, (2 * (h @ w.t())).t()[lo:hi] @ h, atol=1e-4, rtol=1e-4)
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write(f"{lo} {hi}")
    dist.destroy_process_group()


@pytest.mark.parametrize("vocab,world", [(128, 2), (100, 2), (50, 3)])
def test_vocab_parallel_loss_and_embedding_match_full_vocab(vocab, world):
    d = tempfile.mkdtemp(prefix="areal_vp_")
    mp.spawn(_rank_main, args=(world, _free_port(), vocab, d), nprocs=world, join=True)
    bounds = [tuple(int(x) for x in open(os.path.join(d, f"ok{r}")).read().split()) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == vocab
    assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert all((hi - lo) % 8 == 0 for lo, hi in bounds[:-1])  # aligned logit rows on every full shard


def test_shard_ranges_cover_real_vocabularies():
    from areal_vllm_amd.ops import vocab_shard_range

    for vocab in (151936, 152064, 50257, 32000, 128256):
        for tp in (1, 2, 4, 8):
            r = [vocab_shard_range(vocab, tp, i) for i in range(tp)]
            assert r[0][0] == 0 and r[-1][1] == vocab and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    with pytest.raises(AssertionError):
        vocab_shard_range(16, 4, 3)  # 8-row shards: ranks 2 and 3 would be empty


def _critic_worker(rank, world, port, ckpt_in, ckpt_out, tp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank), GLOO_SOCKET_IFNAME="lo")
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from areal_vllm_amd.api.alloc_mode import ParallelStrategy
    from areal_vllm_amd.api.cli_args import MicroBatchSpec, OptimizerConfig, TrainEngineConfig
    from areal_vllm_amd.api.io_struct import FinetuneSpec, SaveLoadMeta
    from areal_vllm_amd.engine.train_engine import NativeTrainEngine

    cfg = TrainEngineConfig(experiment_name="t", trial_name="t", path=ckpt_in or "tiny", init_from_scratch=ckpt_in is None,
                            dtype="float32", gradient_checkpointing=False,
                            optimizer=OptimizerConfig(lr=1e-2, weight_decay=0.0, gradient_clipping=1.0,
                                                      warmup_steps_proportion=0.0, lr_scheduler_type="constant"),
                            mb_spec=MicroBatchSpec(n_mbs=1, max_tokens_per_mb=4096))
    eng = NativeTrainEngine(cfg)
    eng.is_critic = True
    eng.create_process_group(ParallelStrategy(tensor_parallel_size=tp))
    eng.initialize(None, FinetuneSpec(1, 8, 4))
    if ckpt_in is not None:
        if tp > 1:  # the embedding is sharded, the [1, H] value head is not
            assert eng.model.embed.shape[0] == eng.model_config.vocab_size // tp and eng.model.head.shape[0] == 1
        g = torch.Generator().manual_seed(3)
        ids = torch.randint(2, eng.model_config.vocab_size, (4, 24), generator=g)
        batch = dict(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool))

        def loss(proxy, mb):
            return (proxy.values() - 0.5).square().mean()

        eng.train_batch(batch, loss, lambda mb: torch.tensor(float(mb["input_ids"].numel())))
    eng.save(SaveLoadMeta(path=ckpt_out, weight_format="hf", with_optim=False))
    eng.destroy()


def test_critic_under_tensor_parallel_shards_embedding_but_not_value_head():
    from safetensors.torch import load_file

    d = tempfile.mkdtemp(prefix="areal_vpc_")
    init, ref, out = (os.path.join(d, x) for x in ("init", "ref", "tp2"))
    mp.spawn(_critic_worker, args=(1, _free_port(), None, init, 1), nprocs=1, join=True)
    mp.spawn(_critic_worker, args=(1, _free_port(), init, ref, 1), nprocs=1, join=True)
    mp.spawn(_critic_worker, args=(2, _free_port(), init, out, 2), nprocs=2, join=True)
    a, b = (load_file(os.path.join(p, "model.safetensors")) for p in (ref, out))
    assert set(a) == set(b) and "score.weight" in a
    moved = load_file(os.path.join(init, "model.safetensors"))
    assert not torch.equal(moved["model.embed_tokens.weight"], a["model.embed_tokens.weight"])  # the step did train
    for k in a:
        torch.testing.assert_close(b[k], a[k], atol=2e-5, rtol=1e-4, msg=lambda m: f"{k}: {m}")
