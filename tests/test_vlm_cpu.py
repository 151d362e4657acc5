"""Vision-language path end to end on CPU: the rollout engine generates for prompts with images (salted
prefix ids, vision features scattered into the prompt, 3-D rotary positions, position delta during decode)
and the train engine's recomputed log-probs on the same sequences agree with the engine's; a training
step on the multimodal batch moves the vision tower too."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IMG, VS, VE = 500, 502, 503


def _prompt(n_text, grid, seed):
    g = torch.Generator().manual_seed(seed)
    t, h, w = grid
    txt = torch.randint(3, 480, (n_text + 3,), generator=g).tolist()
    ids = txt[:n_text] + [VS] + [IMG] * (t * h * w // 4) + [VE] + txt[n_text:]
    px = torch.randn(t * h * w, 3 * 2 * 14 * 14, generator=g)
    return ids, dict(pixel_values=px, image_grid_thw=torch.tensor([list(grid)]))


def test_vl_rollout_matches_trainer_and_trains():
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", AREAL_PLATFORM="cpu",
                      GLOO_SOCKET_IFNAME="lo")
    from areal_vllm_amd.api.cli_args import InferenceServerConfig, MicroBatchSpec, OptimizerConfig, TrainEngineConfig
    from areal_vllm_amd.api.io_struct import FinetuneSpec
    from areal_vllm_amd.engine.train_engine import NativeTrainEngine
    from areal_vllm_amd.inference.engine import NativeInferenceEngine
    from areal_vllm_amd.utils.data import concat_padded_tensors

    cfg = TrainEngineConfig(experiment_name="v", trial_name="v", path="tiny-vl", init_from_scratch=True, dtype="float32",
                            gradient_checkpointing=False, optimizer=OptimizerConfig(lr=1e-2, gradient_clipping=1.0),
                            mb_spec=MicroBatchSpec(max_tokens_per_mb=256))
    eng = NativeTrainEngine(cfg)
    eng.create_process_group()
    eng.initialize(None, FinetuneSpec(1, 8, 4))
    assert any(n.startswith("visual.") for n, _ in eng.model.named_parameters())
    srv = InferenceServerConfig(model_path="tiny-vl", dtype="float32", context_length=256, max_running_requests=8,
                                num_continuous_decode_steps=2, random_seed=5)
    roll = NativeInferenceEngine(srv, model_config=eng.model_config, device=torch.device("cpu"), num_pages=64)
    roll.initialize(load_weights=False)
    roll.load_weights_from_trainer(eng.model, version=0)
    try:
        cases = [_prompt(5, (1, 4, 6), 1), _prompt(9, (1, 6, 4), 2), _prompt(4, (1, 4, 4), 3)]
        cases.append(cases[0])  # same text and same image again: may share the cached prefix
        text_only = [7, 8, 9, 10, 11]
        done = {}
        import threading
        ev = threading.Event()

        def cb(i):
            def f(res):
                done[i] = res
                if len(done) == len(cases) + 1:
                    ev.set()
            return f

        for i, (ids, mm) in enumerate(cases):
            roll.submit(f"r{i}", ids, max_new_tokens=12, temperature=1.0, callback=cb(i), multimodal=mm)
        roll.submit("rt", text_only, max_new_tokens=12, temperature=1.0, callback=cb(len(cases)))
        assert ev.wait(120)
        assert roll.runner.mm == {} and roll._mm_ids == {}  # image state released with the requests
        rows, behav = [], []
        for i, (ids, mm) in enumerate(cases + [(text_only, None)]):
            r = done[i]
            assert len(r.output_tokens) == 12 and all(0 <= t < 512 for t in r.output_tokens)
            seq = ids + r.output_tokens
            rows.append(dict(input_ids=torch.tensor(seq).unsqueeze(0),
                             attention_mask=torch.ones(1, len(seq), dtype=torch.bool),
                             loss_mask=torch.tensor([0] * len(ids) + [1] * 12).unsqueeze(0),
                             multi_modal_input=[mm]))
            behav.append(r.output_logprobs)
        batch = concat_padded_tensors(rows)
        lp = eng.forward(batch, post_hook=lambda p, mb: p.logprobs(torch.roll(mb["input_ids"], -1, 0)))
        for b, ((ids, _), bl) in enumerate(zip(cases + [(text_only, None)], behav)):
            got = lp[b, len(ids) - 1:len(ids) - 1 + 12]
            torch.testing.assert_close(got, torch.tensor(bl), atol=2e-3, rtol=1e-3)
        # a different image under the same text must not reuse the first request's cached prefix
        other = dict(cases[0][1], pixel_values=cases[0][1]["pixel_values"] + 1.0)
        grid = cases[0][1]["image_grid_thw"]
        assert roll.runner.salted_prompt(cases[0][0], other["pixel_values"], grid) != roll.runner.salted_prompt(
            cases[0][0], cases[0][1]["pixel_values"], grid)
        # a placeholder id sampled by the policy after the image is ordinary text, for both engines
        from areal_vllm_amd.models.vision import mrope_position_ids
        stray = cases[0][0] + [IMG, 7, IMG]
        p3, _, msk = mrope_position_ids(stray, grid.tolist(), [IMG], 2, True)
        assert int(msk.sum()) == 6 and not bool(msk[-1]) and p3[:, -1].tolist() == [int(p3[0, -2]) + 1] * 3

        before = {n: p.detach().clone() for n, p in eng.model.named_parameters() if n.startswith("visual.blocks.0.attn")}

        def loss(proxy, mb):
            l = proxy.logprobs(torch.roll(mb["input_ids"], -1, 0))
            m = mb["loss_mask"].float()
            return -(l * m).sum() / m.sum()

        st = eng.train_batch(batch, loss, lambda mb: mb["loss_mask"].float().sum())
        assert st["update_successful"] == 1.0 and st["grad_norm"] > 0
        after = dict(eng.model.named_parameters())
        assert any(not torch.equal(before[n], after[n]) for n in before)  # the tower is trained
    finally:
        roll.destroy()
        eng.destroy()


def test_named_weight_update_reaches_the_vision_tower():
    """Weight updates that travel by name (collective push, tensor-parallel colocated load) must update the
    vision tower too, and must not drop a tensor silently."""
    import pytest

    from areal_vllm_amd.inference.engine import assign_hf_named_tensors
    from areal_vllm_amd.models import hf_io
    from areal_vllm_amd.models.config import preset
    from areal_vllm_amd.models.transformer import Transformer

    cfg = preset("tiny-vl")
    src = Transformer(cfg, dtype=torch.float32, device="cpu")
    src.flatten()
    src.init_weights(seed=11)
    dst = Transformer(cfg, dtype=torch.float32, device="cpu")
    dst.flatten()
    dst.init_weights(seed=12)
    state = dict(hf_io.native_to_hf_state(src, gather=False))
    assert any(n.startswith("visual.") for n in state)
    assign_hf_named_tensors(dst, state)
    for (n, a), (_, b) in zip(src.named_parameters(), dst.named_parameters()):
        assert torch.equal(a, b), n
    with pytest.raises(KeyError):
        assign_hf_named_tensors(dst, dict(state, **{"model.layers.0.mlp.not_a_weight": torch.zeros(1)}))
