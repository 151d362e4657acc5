"""GPT-2 (LayerNorm with bias, learned positions, ungated MLP with biases) through the rollout engine: the family
the reference converts in ``realhf/api/from_hf/gpt2.py`` and generates with through its own model code."""
import os
import socket

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gpt2_rollout_scores_like_the_trainer_and_takes_in_process_weight_updates():
    from areal_vllm_amd.api.cli_args import (AdvNormConfig, GenerationHyperparameters, InferenceEngineConfig,
                                             InferenceServerConfig, MicroBatchSpec, OptimizerConfig, PPOActorConfig)
    from areal_vllm_amd.api.io_struct import FinetuneSpec, WeightUpdateMeta
    from areal_vllm_amd.engine.local_engine import LocalInferenceEngine
    from areal_vllm_amd.engine.ppo.actor import NativePPOActor
    from areal_vllm_amd.inference.engine import NativeInferenceEngine
    from areal_vllm_amd.workflow.rlvr import RLVRWorkflow

    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                      AREAL_PLATFORM="cpu", GLOO_SOCKET_IFNAME="lo")
    cfg = PPOActorConfig(experiment_name="gpt2", trial_name="t", path="tiny-gpt2", init_from_scratch=True,
                         dtype="float32", optimizer=OptimizerConfig(lr=1e-3), group_size=2,
                         mb_spec=MicroBatchSpec(max_tokens_per_mb=2048), ppo_n_minibatches=1, use_decoupled_loss=True,
                         recompute_logprob=True, kl_ctl=0.0, adv_norm=AdvNormConfig("group", "group", 2),
                         gradient_checkpointing=False)
    actor = NativePPOActor(cfg)
    actor.create_process_group()
    actor.initialize(None, FinetuneSpec(1, 16, 4))
    eng = ro = None
    try:
        assert actor.model_config.family == "gpt2" and actor.model_config.pos_embed == "learned"
        torch.manual_seed(0)
        with torch.no_grad():
            actor.model.flat.flat.normal_(0, 0.08)  # biases and norm biases non-zero
        srv = InferenceServerConfig(model_path="tiny-gpt2", dtype="float32", context_length=256,
                                    max_running_requests=8, num_continuous_decode_steps=3, random_seed=3)
        eng = NativeInferenceEngine(srv, model_config=actor.model_config, device=torch.device("cpu"), num_pages=64)
        eng.initialize(load_weights=False)
        actor.connect_local_engine(eng)
        ro = LocalInferenceEngine(InferenceEngineConfig(max_concurrent_rollouts=8, consumer_batch_size=4,
                                                        max_head_offpolicyness=4), eng)
        ro.initialize(train_data_parallel_size=1)
        meta = WeightUpdateMeta.from_inproc()
        wf = RLVRWorkflow(lambda p, c, pi, ci, **kw: 1.0,
                          GenerationHyperparameters(n_samples=2, max_new_tokens=24, temperature=1.0), reward_in_process=True)
        prompts = [dict(input_ids=[5, 6, 7, 8, 9, 10, 11]), dict(input_ids=[40, 41, 42])]

        def gap():
            b = ro.rollout_batch(prompts, workflow=wf)
            prox = actor.compute_logp(b)
            lm = torch.roll(b["loss_mask"].bool(), -1, 1)
            lm[:, -1] = False
            assert int(lm.sum()) >= 4 * 20
            return float((prox - torch.roll(b["logprobs"], -1, 1)).abs()[lm].max())

        assert gap() > 1e-2  # the engine still holds its own random weights
        for version in (1, 2):
            ro.pause()
            actor.upload_weights(meta)
            actor.set_version(version)
            ro.set_version(version)
            ro.resume()
            assert gap() < 1e-3
            with torch.no_grad():
                actor.model.flat.flat.mul_(0.9)
    finally:
        if ro is not None:
            ro.destroy()
        if eng is not None:
            eng.destroy()
        actor.destroy()


@pytest.mark.parametrize("tp", [1, 2])
def test_gpt2_named_weight_update(tp):
    """Weight updates that travel by checkpoint name (collective push, tensor-parallel colocated load): Conv1D
    layouts, the fused c_attn and this rank's slices of the tensor-parallel projections."""
    from areal_vllm_amd.inference.engine import assign_hf_named_tensors
    from areal_vllm_amd.models import hf_io
    from areal_vllm_amd.models.config import preset
    from areal_vllm_amd.models.transformer import Transformer

    cfg = preset("tiny-gpt2")
    src = Transformer(cfg, dtype=torch.float32, device="cpu")
    src.flatten()
    torch.manual_seed(3)
    with torch.no_grad():
        src.flat.flat.normal_(0, 0.1)
    state = dict(hf_io.native_to_hf_state(src, gather=False))
    for tr in range(tp):
        dst = Transformer(cfg, dtype=torch.float32, device="cpu", tp_size=tp, tp_rank=tr)
        dst.flatten()
        dst.init_weights(seed=12)
        names = list(state)
        assign_hf_named_tensors(dst, {n: state[n] for n in names[: len(names) // 2]})  # updates arrive in chunks
        assign_hf_named_tensors(dst, {n: state[n] for n in names[len(names) // 2:]})
        back = dict(hf_io.native_to_hf_state(dst, gather=False))
        if tp == 1:
            for n in state:
                assert torch.equal(state[n], back[n]), n
        else:  # a shard's own view of the checkpoint: compare what the shard holds against the slices of the source
            D, H, q = cfg.head_dim, cfg.hidden_size, cfg.n_q_heads // tp * cfg.head_dim
            f = cfg.intermediate_size // tp
            w = state["transformer.h.1.attn.c_attn.weight"]
            want = torch.cat([w[:, o + tr * q:o + (tr + 1) * q] for o in (0, H, 2 * H)], 1).t()
            assert torch.equal(dict(dst.named_parameters())["layers.1.attn.qkv"].data, want)
            assert torch.equal(dict(dst.named_parameters())["layers.0.mlp.proj"].data,
                               state["transformer.h.0.mlp.c_proj.weight"][tr * f:(tr + 1) * f].t())
            assert torch.equal(dict(dst.named_parameters())["layers.0.attn.o_bias"].data,
                               state["transformer.h.0.attn.c_proj.bias"])
        with pytest.raises(KeyError):
            assign_hf_named_tensors(dst, {"transformer.h.0.mlp.not_a_weight": torch.zeros(1)})
