"""Peer-memory weight updates between DIFFERENT parallel layouts, on the CPU platform (exported buffers are POSIX
shared memory there): every trainer rank exports the shard it holds - a tensor-parallel slice, a pipeline stage, a
ZeRO-3 shard - and a rollout engine pulls exactly the element ranges that make up its own parameters. Replaces the
reference's all-gather + rank-0 broadcast (``areal/engine/fsdp_engine.py:399-433``); the layouts are those of
``realhf/tests/comm/test_param_realloc.py``."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _trainer_worker(rank, world, port, layout, out_dir, gpu=False):
    """`gpu`: all ranks (and the server) share device 0 - exported buffers are hipIpc handles, the server's interval
    kernel reads the mapped peer memory; the trainer's collectives are staged through host memory (RCCL cannot put two
    ranks on one device). Used by ``tests/test_engine_gpu.py``."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0" if gpu else str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    if gpu:
        os.environ.update(AREAL_HOST_STAGED_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    else:
        os.environ["AREAL_PLATFORM"] = "cpu"
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    import torch.distributed as dist

    from areal_vllm_amd.api.alloc_mode import ParallelStrategy
    from areal_vllm_amd.api.cli_args import InferenceServerConfig, MicroBatchSpec, OptimizerConfig, TrainEngineConfig
    from areal_vllm_amd.api.io_struct import FinetuneSpec, SaveLoadMeta
    from areal_vllm_amd.engine.train_engine import NativeTrainEngine
    from areal_vllm_amd.inference.engine import NativeInferenceEngine
    from areal_vllm_amd.models import hf_io
    from areal_vllm_amd.models.transformer import Transformer

    layout = dict(layout)
    preset_name = layout.pop("_preset", "tiny")
    dtype = "bfloat16" if gpu else "float32"
    cfg = TrainEngineConfig(experiment_name="t", trial_name="pull", path=preset_name, init_from_scratch=True,
                            dtype=dtype,
                            optimizer=OptimizerConfig(lr=1e-2), mb_spec=MicroBatchSpec(max_tokens_per_mb=2048))
    cfg.native.zero_stage = layout.pop("_zero", 1)
    eng = NativeTrainEngine(cfg)
    eng.create_process_group(ParallelStrategy(**layout))
    eng.initialize(None, FinetuneSpec(1, 64, 8))
    if eng.model.flat_experts is not None:
        # equal seeds give the expert-parallel ranks equal expert values: make every rank's experts its own, so that
        # a pull from the wrong rank shows (each expert shard has one holder in the layouts below)
        eng.model.flat_experts.flat.add_(0.01 * (rank + 1))
    meta = eng.export_weight_buffers()
    assert len(meta["ipc_buffers"]) == world and meta["ipc_src"]["zero3"] == (world if cfg.native.zero_stage == 3 else 1)
    eng._ipc_export.refresh()
    if eng._ipc_export_experts is not None:
        assert meta["ipc_src"]["ep"] == eng.grid.ep and len(meta["ipc_src"]["expert_buffers"]) == world
        eng._ipc_export_experts.refresh()
    ckpt = os.path.join(out_dir, "hf")
    eng.save(SaveLoadMeta(path=ckpt, weight_format="hf", with_optim=False))
    dist.barrier()
    if rank == 0:
        srv = NativeInferenceEngine(InferenceServerConfig(model_path=preset_name, dtype=dtype, context_length=128,
                                                          max_running_requests=4),
                                    device=torch.device("cuda:0" if gpu else "cpu"),
                                    num_pages=16).initialize(load_weights=False)
        info = srv.update_weights_from_ipc(None, "", 0, version=3, buffers=meta["ipc_buffers"], src=meta["ipc_src"])
        assert srv.get_version() == 3 and info["bytes"] > 0
        ref = Transformer(eng.model_config, dtype=srv.model.flat.flat.dtype, device="cpu")
        ref.flatten()
        hf_io.load_hf_weights(ref, ckpt)
        got = srv.model.flat.flat.cpu()
        for name, (o, numel) in ref.flat.offsets.items():
            assert torch.equal(got[o:o + numel], ref.flat.flat[o:o + numel]), name
        srv.destroy()
    dist.barrier()
    eng.destroy()


@pytest.mark.parametrize("name,world,layout", [
    ("tp2", 2, dict(tensor_parallel_size=2)),
    ("pp2", 2, dict(pipeline_parallel_size=2)),
    ("zero3_dp2", 2, dict(data_parallel_size=2, _zero=3)),
    ("dp2", 2, dict(data_parallel_size=2)),
    ("tp2pp2", 4, dict(tensor_parallel_size=2, pipeline_parallel_size=2)),
    # mixture-of-experts trainers: experts inside the dense buffers of two pipeline stages; expert-parallel ranks
    # that export their experts as a second buffer (split over tp, and whole inside a tensor-parallel group)
    ("moe_pp2", 2, dict(pipeline_parallel_size=2, _preset="tiny-moe")),
    ("moe_dp2ep2", 2, dict(data_parallel_size=2, expert_parallel_size=2, _preset="tiny-moe")),
    ("moe_dp2tp2ep2", 4, dict(data_parallel_size=2, tensor_parallel_size=2, expert_parallel_size=2, _preset="tiny-moe")),
    ("moe_tp2ep2etp1", 2, dict(tensor_parallel_size=2, expert_parallel_size=2, expert_tensor_parallel_size=1,
                               _preset="tiny-moe")),
])
def test_unsharded_server_pulls_from_a_sharded_trainer(name, world, layout):
    out = tempfile.mkdtemp(prefix=f"areal_pull_{name}_")
    mp.spawn(_trainer_worker, args=(world, _free_port(), layout, out), nprocs=world, join=True)


def test_zero3_shard_map_matches_the_sharder():
    """The layout arithmetic servers use to find a parameter's elements in ZeRO-3 shards against the shards the
    sharder itself cuts (group of one rank here: offsets and padding; the multi-rank split is covered above)."""
    sys.path.insert(0, ROOT)
    from areal_vllm_amd.models.config import preset
    from areal_vllm_amd.models.transformer import Transformer
    from areal_vllm_amd.parallel.zero3 import Zero3, shard_map

    m = Transformer(preset("tiny"), dtype=torch.float32, device="cpu")
    m.flatten()
    m.init_weights(seed=2)
    want = {n: p.detach().clone() for n, p in m.named_parameters()}
    offsets = dict(m.flat.offsets)
    sm = shard_map(m, 1)
    z = Zero3(m, None, group=None)
    assert sm.shard_numel == z.shard_numel and sm.unit_numel == [u.numel for u in z.units]
    for name, (o, numel) in offsets.items():
        tri = sm.translate(np.asarray([[o, o + numel]]), np.asarray([[0, numel]]))[0]
        got = torch.cat([z.param_shard[s:s + ln] for s, _d, ln in tri.tolist()])
        assert torch.equal(got, want[name].flatten()), name
