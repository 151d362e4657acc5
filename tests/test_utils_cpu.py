"""Unit tests of the utility layer: packing, micro-batching, name_resolve, stats tracker, frequency
control, functional RL math, data loader, reward parsing, OpenAI-style client."""
import asyncio
import os
import threading
import time

import numpy as np
import pytest
import torch

from areal_vllm_amd.api.cli_args import AdvNormConfig, MicroBatchSpec
from areal_vllm_amd.utils import datapack, name_resolve, stats_tracker
from areal_vllm_amd.utils.data import (
    concat_padded_tensors,
    pad_mb_list,
    pad_sequences_to_tensors,
    pack_tensor_dict,
    split_padded_tensor_dict_into_mb_list,
    unpack_sequence,
)


def _padded(lens, vocab=100, seed=0):
    g = torch.Generator().manual_seed(seed)
    rows = [dict(input_ids=torch.randint(1, vocab, (n,), generator=g), loss_mask=torch.ones(n, dtype=torch.bool))
            for n in lens]
    d = pad_sequences_to_tensors(rows)
    d["rewards"] = torch.arange(len(lens)).float()
    return d


# ---------------------------------------------------------------------------- data packing
@pytest.mark.parametrize("n_mbs,max_tokens", [(1, None), (3, None), (1, 64), (2, 48)])
def test_micro_batch_split_roundtrip(n_mbs, max_tokens):
    lens = [5, 17, 30, 9, 22, 41, 3, 12]
    data = _padded(lens)
    mbl = split_padded_tensor_dict_into_mb_list(data, MicroBatchSpec(n_mbs=n_mbs, max_tokens_per_mb=max_tokens))
    assert len(mbl.mbs) >= n_mbs
    if max_tokens is not None:
        assert all(t <= max_tokens for t in mbl.group_lens)
    assert sorted(mbl.forward_indices) == list(range(len(lens)))
    # forward order followed by backward_indices restores the input order
    ids_fwd = [data["input_ids"][i] for i in mbl.forward_indices]
    for b, j in enumerate(mbl.backward_indices):
        assert torch.equal(ids_fwd[j], data["input_ids"][b])
    padded = pad_mb_list(mbl, align_to=4)
    for pmb, pad, to in zip(padded.padded_mbs, padded.padding_lengths, padded.padded_to_lengths):
        assert pmb["input_ids"].shape[0] == to and to % 4 == 0
        assert int(pmb["cu_seqlens"][-1]) == to


def test_pack_unpack_and_concat():
    data = _padded([4, 9, 2])
    packed = pack_tensor_dict(dict(data))
    assert packed["cu_seqlens"].tolist() == [0, 4, 13, 15] and packed["max_seqlen"] == 9
    seqs = unpack_sequence(packed["input_ids"], cu_seqlens=packed["cu_seqlens"])
    assert [len(s) for s in seqs] == [4, 9, 2]
    both = concat_padded_tensors([_padded([3, 5]), _padded([11])])
    assert both["input_ids"].shape == (3, 11) and both["attention_mask"].sum(-1).tolist() == [3, 5, 11]


def test_ffd_and_balanced_partition():
    lens = [7, 3, 9, 1, 4, 8, 2, 6]
    groups = datapack.ffd_allocate(lens, 10, 2)
    assert sorted(i for g in groups for i in g) == list(range(8))
    assert all(sum(lens[i] for i in g) <= 10 for g in groups)
    bounds = datapack.partition_balanced(lens, 3)
    assert len(bounds) == 4 and bounds[0] == 0 and bounds[-1] == 8


# ----------------------------------------------------------------------------- name_resolve
@pytest.mark.parametrize("kind", ["memory", "nfs"])
def test_name_resolve_contract(kind, tmp_path):
    repo = (name_resolve.MemoryNameRecordRepository() if kind == "memory"
            else name_resolve.NfsNameRecordRepository(str(tmp_path)))
    repo.add("a/b/c", "1")
    with pytest.raises(name_resolve.NameEntryExistsError):
        repo.add("a/b/c", "2")
    repo.add("a/b/c", "2", replace=True)
    assert repo.get("a/b/c") == "2"
    repo.add_subentry("a/servers", "h1:1")
    repo.add_subentry("a/servers", "h2:2")
    assert sorted(repo.get_subtree("a/servers")) == ["h1:1", "h2:2"]
    assert len(repo.find_subtree("a")) == 3
    with pytest.raises(name_resolve.NameEntryNotFoundError):
        repo.get("missing")
    with pytest.raises(TimeoutError):
        repo.wait("missing", timeout=0.2, poll_frequency=0.05)
    threading.Timer(0.2, lambda: repo.add("late", "x")).start()
    assert repo.wait("late", timeout=5, poll_frequency=0.05) == "x"
    repo.delete("a/b/c")
    with pytest.raises(name_resolve.NameEntryNotFoundError):
        repo.delete("a/b/c")
    repo.clear_subtree("a")
    assert repo.get_subtree("a") == []
    repo.reset()


# ---------------------------------------------------------------------------- stats tracker
def test_stats_tracker_reductions_and_scopes():
    t = stats_tracker.DistributedStatsTracker()
    mask = torch.tensor([[1, 1, 0], [1, 0, 0]], dtype=torch.bool)
    vals = torch.tensor([[1.0, 3.0, 100.0], [5.0, 100.0, 100.0]])
    with t.scope("ppo"):
        t.denominator(valid=mask)
        t.stat(denominator="valid", x=vals)
        t.scalar(lr=0.1)
        t.scalar(lr=0.3)
    out = t.export()
    assert out["ppo/x/avg"] == pytest.approx(3.0) and out["ppo/x/min"] == 1.0 and out["ppo/x/max"] == 5.0
    assert out["ppo/valid"] == 3 and out["ppo/lr"] == pytest.approx(0.2)
    assert t.export() == {}
    with t.record_timing("step"):
        time.sleep(0.01)
    assert t.export()["timeperf/step"] >= 0.01


def test_stats_tracker_rejects_shape_mismatch():
    t = stats_tracker.DistributedStatsTracker()
    t.denominator(m=torch.ones(2, 3, dtype=torch.bool))
    with pytest.raises(Exception):
        t.stat(denominator="m", x=torch.zeros(2, 4))
    with pytest.raises(Exception):
        t.denominator(bad=torch.ones(3))  # not boolean


# -------------------------------------------------------------------------------- timeutil
def test_frequency_controls():
    from areal_vllm_amd.utils.timeutil import EpochStepTimeFreqCtl, FrequencyControl

    f = FrequencyControl(frequency_steps=3)
    assert [f.check() for _ in range(7)] == [False, False, True, False, False, True, False]
    g = FrequencyControl(frequency_seconds=0.05)
    assert not g.check()
    time.sleep(0.06)
    assert g.check() and not g.check()
    e = EpochStepTimeFreqCtl(freq_epoch=1, freq_step=None, freq_sec=None)
    assert not e.check(epochs=0, steps=1) and e.check(epochs=1, steps=1)
    sd = e.state_dict()
    e2 = EpochStepTimeFreqCtl(freq_epoch=1, freq_step=None, freq_sec=None)
    e2.load_state_dict(sd)
    assert e2.state_dict()["epoch_ctl"]["steps"] == sd["epoch_ctl"]["steps"]


# ------------------------------------------------------------------------------ data loader
def test_stateful_dataloader_resumes_mid_epoch():
    from areal_vllm_amd.utils.dataloader import StatefulDataLoader

    ds = list(range(20))
    a = StatefulDataLoader(ds, batch_size=4, shuffle=True, seed=3)
    it = iter(a)
    first = [next(it), next(it)]
    sd = a.state_dict()
    rest_a = list(it)
    b = StatefulDataLoader(ds, batch_size=4, shuffle=True, seed=3)
    b.load_state_dict(sd)
    assert list(b) == rest_a and len(rest_a) == 3
    assert sorted(x for bt in first + rest_a for x in bt) == ds
    # next epoch reshuffles
    assert list(b) != first + rest_a


# ------------------------------------------------------------------------------- functional
def test_adv_norm_levels():
    from areal_vllm_amd.engine.ppo.actor import AdvNorm

    torch.manual_seed(0)
    adv = torch.randn(8, 6) * 3 + 1
    mask = torch.ones(8, 6)
    out = AdvNorm(AdvNormConfig(mean_level="batch", std_level="batch"))(adv, mask)
    assert out.mean().abs() < 1e-5 and (out.std(unbiased=False) - 1).abs() < 1e-3
    out = AdvNorm(AdvNormConfig(mean_level="group", std_level="group", group_size=4))(adv, mask)
    g = out.view(2, 4 * 6)
    assert g.mean(-1).abs().max() < 1e-5 and (g.std(-1, unbiased=False) - 1).abs().max() < 1e-3
    out = AdvNorm(AdvNormConfig(mean_level="group", std_level="batch", group_size=4))(adv, mask)
    assert out.view(2, -1).mean(-1).abs().max() < 1e-5
    out = AdvNorm(AdvNormConfig(mean_level="none", std_level="none"))(adv, mask)
    assert torch.equal(out, adv)
    # masked positions do not influence the statistics of the valid ones
    m2 = mask.clone()
    m2[:, 3:] = 0
    adv2 = adv.clone()
    adv2[:, 3:] = 1e6
    out = AdvNorm(AdvNormConfig(mean_level="batch", std_level="batch"))(adv2, m2)
    v = adv[:, :3]
    torch.testing.assert_close(out[:, :3], (v - v.mean()) / (v.std(unbiased=False) + 1e-5), atol=1e-4, rtol=1e-4)
    with pytest.raises(Exception):
        AdvNorm(AdvNormConfig(mean_level="group", std_level="group", group_size=None))


def test_ppo_actor_loss_matches_definition():
    from areal_vllm_amd.utils.functional import ppo_actor_loss_fn

    torch.manual_seed(1)
    T = 50
    logp, old, prox = torch.randn(T) * 0.1 - 1, torch.randn(T) * 0.1 - 1, torch.randn(T) * 0.1 - 1
    adv = torch.randn(T)
    mask = torch.rand(T) > 0.2
    # on-policy form: proximal == behaviour policy
    loss, stat = ppo_actor_loss_fn(logp, old, old, adv, eps_clip=0.2, loss_mask=mask)
    ratio = torch.exp(logp - old)
    ref = torch.max(-adv * ratio, -adv * ratio.clamp(0.8, 1.2))
    assert loss.item() == pytest.approx((ref * mask).sum().item() / mask.sum().item(), rel=1e-5)
    # decoupled: ratio against the proximal policy, weighted by the capped behaviour importance weight
    loss2, stat2 = ppo_actor_loss_fn(logp, prox, old, adv, eps_clip=0.2, loss_mask=mask, behav_imp_weight_cap=1.05)
    r2 = torch.exp(logp - prox)
    l2 = torch.max(-adv * r2, -adv * r2.clamp(0.8, 1.2))
    w = torch.exp(prox - old)
    keep = (w <= 1.05) & mask
    ref2 = (l2 * w * keep).sum() / mask.sum()
    assert loss2.item() == pytest.approx(ref2.item(), rel=1e-5)
    assert stat2["behave_mask"].sum() == keep.sum()
    # asymmetric clipping widens only the upper bound
    loss4, stat4 = ppo_actor_loss_fn(logp + 0.5, old, old, adv.abs(), eps_clip=0.2, loss_mask=mask, eps_clip_higher=0.9)
    r4 = torch.exp(logp + 0.5 - old)
    ref4 = torch.max(-adv.abs() * r4, -adv.abs() * r4.clamp(0.8, 1.9))
    assert loss4.item() == pytest.approx((ref4 * mask).sum().item() / mask.sum().item(), rel=1e-5)
    # dual clip bounds the loss for very negative advantages
    loss3, _ = ppo_actor_loss_fn(logp + 3, old, old, -adv.abs(), eps_clip=0.2, loss_mask=mask, c_clip=3.0)
    assert torch.isfinite(loss3) and loss3.item() <= 3.0 * adv.abs().max().item()


def test_dynamic_sampling_and_overlong_penalty():
    from areal_vllm_amd.utils.functional import dynamic_sampling, reward_overlong_penalty

    data = dict(rewards=torch.tensor([1.0, 1.0, 0.0, 1.0, 0.0, 0.0]), x=torch.arange(6))
    out, stat = dynamic_sampling(data, 2)
    assert out["x"].tolist() == [2, 3] and stat["n_group_filtered"] == 2 and stat["n_group_keeped"] == 1
    d = dict(rewards=torch.tensor([1.0, 1.0]), loss_mask=torch.zeros(2, 20), )
    d["loss_mask"][0, :18] = 1  # 18 response tokens, limit 20, buffer 5 -> 3 tokens over the soft limit
    d["loss_mask"][1, :5] = 1
    reward_overlong_penalty(d, overlong_tokens=5, overlong_penalty_factor=1.0, max_response_length=20)
    assert d["rewards"][0].item() == pytest.approx(1.0 - 3 / 5) and d["rewards"][1].item() == 1.0


def test_gae_reference_matches_loop():
    from areal_vllm_amd import ops

    torch.manual_seed(0)
    B, S = 3, 9
    rewards, values = torch.randn(B, S), torch.randn(B, S)
    no_eos = torch.tensor([True, False, False])
    adv, ret = ops.gae_padded(rewards, values, no_eos, 0.95, 0.9)
    for b in range(B):
        last = 0.0
        for t in reversed(range(S - 1)):
            nxt = values[b, t + 1] if (t + 1 < S - 1 or no_eos[b]) else 0.0
            delta = rewards[b, t] + 0.95 * nxt - values[b, t]
            last = delta + 0.95 * 0.9 * last
            assert adv[b, t].item() == pytest.approx(float(last), abs=1e-4)
    assert torch.allclose(ret[:, :-1], (adv + values)[:, :-1], atol=1e-5)


# ----------------------------------------------------------------------------------- reward
@pytest.mark.parametrize("answer,solution,ok", [
    ("so \\boxed{72}", "Natalia sold 24 clips.\n#### 72", 1),
    ("\\boxed{-\\frac{2}{3}}", "\\boxed{-\\frac{2}{3}}", 1),
    ("\\boxed{\\dfrac{1}{2}}", "0.5", 1),
    ("\\boxed{1,000}", "#### 1000", 1),
    ("\\boxed{2\\sqrt{2}}", "\\boxed{\\sqrt{8}}", 1),
    ("\\boxed{x^2+2x+1}", "\\boxed{(x+1)^2}", 1),
    ("\\boxed{(1,2)}", "\\boxed{(1, 2)}", 1),
    ("\\boxed{[1,2)}", "\\boxed{(1,2)}", 0),
    ("\\boxed{71}", "#### 72", 0),
    ("I think it is 72", "#### 72", 0),
    ("\\boxed{10 \\text{ cm}}", "\\boxed{10}", 1),
])
def test_math_verifier(answer, solution, ok):
    from areal_vllm_amd.reward.math_parser import process_results

    assert process_results(answer, solution)[0] == ok


def test_reward_registry_and_vision_rewards():
    from areal_vllm_amd.reward import get_custom_reward_fn
    from areal_vllm_amd.reward.clevr_count_70k import clevr_count_70k_reward_fn
    from areal_vllm_amd.reward.geometry3k import geometry3k_reward_fn

    assert get_custom_reward_fn("openai/gsm8k")("", "\\boxed{5}", [], [], answer="#### 5") == 1.0
    assert clevr_count_70k_reward_fn("", "I count [3] objects", [], [], answer="3") == 1.0
    assert clevr_count_70k_reward_fn("", "three", [], [], answer="3") == 0.0
    assert geometry3k_reward_fn("", "thus [12.50]", [], [], answer="12.5") == 1.0
    with pytest.raises(ValueError):
        get_custom_reward_fn("unknown/dataset")


def test_async_reward_wrapper_timeout_and_value():
    from areal_vllm_amd.api.reward_api import AsyncRewardWrapper

    w = AsyncRewardWrapper(_slow_reward, timeout_seconds=5.0)  # covers the pool start-up

    async def go():
        fast = await w("p", "c", [1], [2], delay=0.0)
        slow = await w("p", "c", [1], [2], delay=12.0)
        return fast, slow

    fast, slow = asyncio.run(go())
    assert fast == 1.0 and slow == 0


def _slow_reward(prompt, completion, prompt_ids, completion_ids, delay=0.0, **kw):
    time.sleep(delay)
    return 1.0


# ------------------------------------------------------------------- OpenAI-style client
class _Tok:
    eos_token_id, pad_token_id = 1, 0

    def apply_chat_template(self, messages, tokenize=True, add_generation_prompt=True, **kw):
        ids = [2 + len(m["content"]) % 50 for m in messages] + [60]
        return ids if tokenize else " ".join(map(str, ids))

    def decode(self, ids):
        return "tok:" + ",".join(map(str, ids))


class _Eng:
    async def agenerate(self, req):
        from areal_vllm_amd.api.io_struct import ModelResponse

        n = min(3, req.gconfig.max_new_tokens)
        return ModelResponse(input_tokens=list(req.input_ids), output_tokens=[7] * n, output_logprobs=[-0.5] * n,
                             output_versions=[0] * n, stop_reason="stop")


def test_openai_client_reward_backprop():
    from areal_vllm_amd.experimental.openai import ArealOpenAI

    c = ArealOpenAI(_Eng(), _Tok())

    async def go():
        m = [{"role": "user", "content": "q"}]
        a = await c.chat.completions.create(messages=m, max_completion_tokens=8)
        m2 = m + [{"role": "assistant", "content": a.choices[0].message.content}, {"role": "user", "content": "retry"}]
        b = await c.chat.completions.create(messages=m2, max_completion_tokens=8)
        return a, b

    a, b = asyncio.run(go())
    assert a.usage.completion_tokens == 3 and a.choices[0].finish_reason == "stop"
    c.set_reward(b.id, 1.0)
    out = c.export_completions(turn_discount=0.5)
    assert out[b.id].reward == 1.0 and out[a.id].reward == pytest.approx(0.5)
    row = out[a.id].to_tensor_dict()
    assert row["input_ids"].shape == (1, 2 + 3) and row["loss_mask"].sum() == 3
    with pytest.raises(KeyError):
        c.set_reward("nope", 1.0)


@pytest.mark.parametrize("fmt,text", [
    ("qwen25", 'sure <tool_call>\n{"name": "search", "arguments": {"q": "x"}}\n</tool_call>'),
    ("llama3", 'sure <|python_tag|>{"name": "search", "parameters": {"q": "x"}}'),
    ("mistral", 'sure [TOOL_CALLS] [{"name": "search", "arguments": {"q": "x"}}]'),
])
def test_tool_call_parser(fmt, text):
    from areal_vllm_amd.experimental.openai.tool_call_parser import process_tool_calls

    tools = [{"type": "function", "function": {"name": "search", "parameters": {}}}]
    calls, rest, reason = process_tool_calls(text, tools, fmt, "stop")
    assert reason == "tool_calls" and rest == "sure"
    assert calls[0].function.name == "search" and '"q": "x"' in calls[0].function.arguments
    calls, rest, reason = process_tool_calls("plain answer", tools, fmt, "stop")
    assert calls is None and reason == "stop"


def test_wrapper_and_printing():
    from areal_vllm_amd.utils.printing import tabulate_stats
    from areal_vllm_amd.utils.wrapper import wrap, wrap_get_method, wrap_remove_meta, wrapable

    class Src:
        @wrapable()
        def f(self, x):
            return x + 1

        @wrapable(name="g2")
        def g(self, x):
            return x * 2

        def hidden(self):
            return 0

    class Dst:
        pass

    d = Dst()
    wrap(d, Src())
    assert d.f(1) == 2 and d.g2(3) == 6 and not hasattr(d, "hidden")
    d2 = Dst()
    wrap(d2, Src(), transform=lambda *a, **k: ("via", wrap_get_method(k)(*a, **wrap_remove_meta(dict(k)))))
    assert d2.f(1) == ("via", 2)
    assert "a" in tabulate_stats({"a": 1.0, "b": 2.0})


# ------------------------------------------------------------------- tool-agent workflow
class _CharTok:
    """Byte-level tokenizer: ids are byte values shifted by 3 (0 pad, 1 eos)."""
    eos_token_id, pad_token_id = 1, 0

    def encode(self, text):
        return [b + 3 for b in text.encode()]

    def decode(self, ids, **kw):
        return bytes(i - 3 for i in ids if i >= 3).decode(errors="ignore")

    def apply_chat_template(self, messages, tools=None, tokenize=True, add_generation_prompt=True, **kw):
        text = "".join(f"<{m['role']}>{m['content']}" for m in messages) + "<assistant>"
        return self.encode(text) if tokenize else text


class _ScriptedEngine:
    """First turn of an episode calls the search tool, the second one answers from the observation."""

    def __init__(self):
        self.version = 0
        self.seen_rids = []

    def get_version(self):
        return self.version

    async def agenerate(self, req):
        from areal_vllm_amd.api.io_struct import ModelResponse

        tok = _CharTok()
        self.seen_rids.append(req.rid)
        ctx = tok.decode(req.input_ids)
        if "<tool_response>" not in ctx:
            text = 'let me look <tool_call>\n{"name": "search", "arguments": {"query": "capital France"}}\n</tool_call>'
        else:
            text = "<answer>Paris</answer>" if "Paris" in ctx else "<answer>unknown</answer>"
        ids = tok.encode(text)[: req.gconfig.max_new_tokens]
        self.version += 1
        return ModelResponse(input_tokens=list(req.input_ids), output_tokens=ids, output_logprobs=[-0.1] * len(ids),
                             output_versions=[self.version] * len(ids), stop_reason="stop")


def test_tool_agent_workflow_masks_observations_and_scores_answer():
    from areal_vllm_amd.api.cli_args import GenerationHyperparameters
    from areal_vllm_amd.impl.environment.local_search_env import BM25Index, LocalSearchEnv, f1_answer_reward
    from areal_vllm_amd.workflow.tool_agent import ToolAgentWorkflow

    index = BM25Index([{"url": "doc://fr", "title": "France", "text": "Paris is the capital of France."},
                       {"url": "doc://de", "title": "Germany", "text": "Berlin is the capital of Germany."}])
    assert index.search("capital of France")[0] == 0
    eng = _ScriptedEngine()
    keys = ["num_llm_gens", "num_tool_calls", "num_success_search_queries", "score"]
    wf = ToolAgentWorkflow(f1_answer_reward, GenerationHyperparameters(n_samples=2, max_new_tokens=256), _CharTok(),
                           env_factory=lambda: LocalSearchEnv(index), max_turns=4, agent_stats_keys=keys)
    data = dict(messages=[{"role": "user", "content": "What is the capital of France?"}], answer=["Paris"])
    batch = asyncio.run(wf.arun_episode(eng, data))
    assert batch["input_ids"].shape[0] == 2 and torch.all(batch["rewards"] == 1.0)
    tok = _CharTok()
    for b in range(2):
        n = int(batch["attention_mask"][b].sum())
        ids, lm, ver = batch["input_ids"][b, :n].tolist(), batch["loss_mask"][b, :n].bool(), batch["versions"][b, :n]
        trained = tok.decode([t for t, m in zip(ids, lm.tolist()) if m])
        untrained = tok.decode([t for t, m in zip(ids, lm.tolist()) if not m])
        assert "<tool_call>" in trained and "<answer>Paris</answer>" in trained
        assert "<tool_response>" in untrained and "Paris is the capital" in untrained and "tool_call" not in untrained
        assert torch.all(ver[~lm] == -1) and len(set(ver[lm].tolist())) == 2  # two turns, two weight versions
        assert float(batch["num_llm_gens"][b]) == 2 and float(batch["num_tool_calls"][b]) == 1
        assert float(batch["num_success_search_queries"][b]) == 1
    # both turns of an episode share one request id (prefix-cache affinity)
    assert len(set(eng.seen_rids)) == 2 and len(eng.seen_rids) == 4


def test_countdown_dataset_and_reward():
    from areal_vllm_amd.dataset import get_custom_dataset
    from areal_vllm_amd.dataset.countdown import generate
    from areal_vllm_amd.reward import get_custom_reward_fn

    rows = generate(40, seed=3)
    fn = get_custom_reward_fn("data/countdown/train.jsonl")
    for r in rows:
        good = f"<think>..</think><answer> {r['solution']} </answer>"
        assert fn("", good, [], [], numbers=r["numbers"], target=r["target"]) == 1.0
        assert fn("", "<answer> 1 + 1 </answer>", [], [], numbers=r["numbers"], target=r["target"]) == 0.1
        assert fn("", "__import__('os')", [], [], numbers=r["numbers"], target=r["target"]) == 0.0
    d0 = get_custom_dataset("countdown", 0, 2, type="rl", split="train", n_rows=64)
    d1 = get_custom_dataset("countdown", 1, 2, type="rl", split="train", n_rows=64)
    assert len(d0) == len(d1) == 32 and d0[0]["query_id"] != d1[0]["query_id"]


def test_fused_proximal_logp_gives_the_same_update():
    """With one minibatch per update the proximal log-probs taken from the training forward equal the
    separately recomputed ones, so both variants must take the same optimizer step."""
    import os

    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29583", AREAL_PLATFORM="cpu")
    from areal_vllm_amd.api.cli_args import AdvNormConfig, MicroBatchSpec, OptimizerConfig, PPOActorConfig
    from areal_vllm_amd.api.io_struct import FinetuneSpec
    from areal_vllm_amd.engine.ppo.actor import NativePPOActor

    def run(fuse):
        cfg = PPOActorConfig(experiment_name="f", trial_name="f", path="tiny", init_from_scratch=True, dtype="float32",
                             optimizer=OptimizerConfig(lr=1e-3, gradient_clipping=1.0), group_size=2,
                             mb_spec=MicroBatchSpec(max_tokens_per_mb=4096), ppo_n_minibatches=1, use_decoupled_loss=True,
                             recompute_logprob=True, fuse_proximal_logp=fuse, kl_ctl=0.0, gradient_checkpointing=False,
                             adv_norm=AdvNormConfig("group", "group", 2), behav_imp_weight_cap=5.0)
        a = NativePPOActor(cfg)
        a.create_process_group()
        a.initialize(None, FinetuneSpec(1, 8, 4))
        g = torch.Generator().manual_seed(0)
        B, S = 4, 24
        batch = dict(input_ids=torch.randint(2, 500, (B, S), generator=g), attention_mask=torch.ones(B, S, dtype=torch.bool),
                     loss_mask=torch.cat([torch.zeros(B, 8), torch.ones(B, S - 8)], 1).long(),
                     logprobs=-torch.rand(B, S, generator=g) - 5.0, versions=torch.zeros(B, S, dtype=torch.long),
                     rewards=torch.tensor([1.0, 0.0, 0.0, 1.0]))
        if not a.actor.fuses_proximal_logp():
            batch["prox_logp"] = a.compute_logp(batch)
        a.compute_advantages(batch)
        st = a.ppo_update(batch)
        w = a.model.flat.flat.clone()
        a.destroy()
        return st[0]["grad_norm"], w

    g0, w0 = run(False)
    g1, w1 = run(True)
    assert g0 == pytest.approx(g1, rel=1e-5) and torch.allclose(w0, w1, atol=1e-6)


def _harness_body(rank, world, shared):
    import torch.distributed as dist

    from areal_vllm_amd.utils import name_resolve

    t = torch.tensor([float(rank + shared["bias"])])
    dist.all_reduce(t)
    name_resolve.add_subentry("harness/peers", str(rank))
    dist.barrier()
    return float(t), sorted(name_resolve.get_subtree("harness/peers"))


def test_local_multiprocess_harness():
    from areal_vllm_amd.base.testing import LocalMultiProcessTest

    res = LocalMultiProcessTest(3, _harness_body, shared=dict(bias=1)).launch()
    assert all(r[0] == 6.0 and r[1] == ["0", "1", "2"] for r in res)

    def boom(rank, world, shared):
        raise ValueError("x")

    with pytest.raises(Exception):
        LocalMultiProcessTest(1, _harness_body, shared=None).launch()  # shared=None -> TypeError inside the rank


def _mb_sync_body(rank, world, shared):
    from areal_vllm_amd.api.cli_args import MicroBatchSpec
    from areal_vllm_amd.utils.data import allocate_balanced_mbs_synced

    lens = [[100] * 4, [100] * 9, [100] * 3][rank]  # 1, 3 and 1 micro-batches of <= 400 tokens on their own
    out = []
    for _ in range(3):  # repeated calls must stay aligned (no rank may skip a collective)
        groups = allocate_balanced_mbs_synced(MicroBatchSpec(max_tokens_per_mb=400), lens)
        out.append((len(groups), sorted(i for g in groups for i in g) == list(range(len(lens)))))
    return out


def test_micro_batch_count_is_synchronised_across_ranks():
    from areal_vllm_amd.base.testing import LocalMultiProcessTest

    res = LocalMultiProcessTest(3, _mb_sync_body, timeout=120).launch()
    counts = {c for r in res for c, _ in r}
    assert len(counts) == 1 and counts.pop() >= 3 and all(ok for r in res for _, ok in r)


def test_embedding_and_linear_accumulate_into_main_grad():
    """Embedding rows and linear weights add their gradients straight into the fp32 `main_grad` buffer
    (no dense `.grad`), over several micro-batches, and match plain autograd."""
    import torch
    import torch.nn.functional as F

    from areal_vllm_amd import ops

    torch.manual_seed(0)
    V, H, N = 50, 16, 24
    emb = torch.nn.Parameter(torch.randn(V, H))
    w = torch.nn.Parameter(torch.randn(N, H))
    for p in (emb, w):
        p.main_grad = torch.zeros_like(p, dtype=torch.float32)
    ref_e, ref_w = emb.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    for step in range(3):
        ids = torch.randint(0, V, (7, 5))
        ids[0, :3] = 4  # repeated rows accumulate
        g = torch.randn(7, 5, N)
        ops.linear(ops.embedding(ids, emb), w).backward(g)
        F.linear(F.embedding(ids, ref_e), ref_w).backward(g)
    assert emb.grad is None and w.grad is None
    torch.testing.assert_close(emb.main_grad, ref_e.grad)
    torch.testing.assert_close(w.main_grad, ref_w.grad, atol=1e-4, rtol=1e-4)
    # without a main_grad buffer (plain modules, inference) both fall back to the stock ops
    e2 = torch.nn.Parameter(torch.randn(V, H))
    ops.embedding(torch.tensor([1, 2]), e2).sum().backward()
    assert e2.grad is not None
