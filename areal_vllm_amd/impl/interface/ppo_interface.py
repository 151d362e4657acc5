"""PPO actor / critic interfaces of the worker runtime.

The handlers unpack a `SequenceSample` (groups of answers per prompt) into the padded batch layout and
run the same algorithm objects as the SPMD API (`engine.ppo.actor.PPOActor`, `engine.ppo.critic.PPOCritic`):
log-prob recompute, KL-shaped rewards + GAE kernel + advantage normalisation, minibatched decoupled-PPO
update, value clipping. On top they report how far off-policy a batch is (weight version when each answer
started / finished vs. the trainer's version). Behaviour of
``realhf/impl/model/interface/ppo_interface.py:210-913`` (actor) and ``:984-1341`` (critic).
"""
import dataclasses
from typing import Any, Dict, List, Optional

import torch

from ...api.cli_args import AdvNormConfig, GenerationHyperparameters, InferenceServerConfig, PPOActorConfig, \
    PPOCriticConfig
from ...engine.ppo.actor import PPOActor
from ...engine.ppo.critic import PPOCritic
from ...system.data_api import SequenceSample
from ...system.model_api import Model, ModelInterface, register_interface
from ...utils import logging, stats_tracker
from ._convert import group_sizes, pad_sequences, result_sample, seqlens_of, short_from_padded, to_padded_batch

logger = logging.getLogger("ppo-interface")


def _actor_config(model: Model, **kw) -> PPOActorConfig:
    base = model.module.config
    fields = {f.name for f in dataclasses.fields(PPOActorConfig)}
    d = {f.name: getattr(base, f.name) for f in dataclasses.fields(type(base)) if f.name in fields}
    d.update({k: v for k, v in kw.items() if k in fields})
    return PPOActorConfig(**d)


def _padded_rl_batch(data: SequenceSample, device) -> Dict[str, torch.Tensor]:
    lens = seqlens_of(data)
    S = max(lens)
    batch = to_padded_batch(data, device, short_keys=("packed_logprobs",),
                            seq_keys=("rewards", "seq_no_eos_mask", "version_start", "version_end"),
                            token_keys=("prompt_mask", "values"))
    pm = batch.pop("prompt_mask", None)
    am = batch["attention_mask"]
    batch["loss_mask"] = (am & ~pm.bool()) if pm is not None else am.clone()
    if "packed_logprobs" in batch:
        batch["logprobs"] = batch.pop("packed_logprobs")
    else:
        batch["logprobs"] = torch.zeros(len(lens), S, device=device)
    short = [n - 1 for n in lens]
    for src, dst in (("proximal_logprobs", "prox_logp"), ("packed_ref_logprobs", "ref_logp")):
        if src in data.keys and data.data.get(src) is not None:
            batch[dst] = pad_sequences(data.data[src].to(device).float(), short, S)
    if "rewards" in batch:
        batch["rewards"] = batch["rewards"].float()
    return batch


class PPOActorInterface(ModelInterface):
    def __init__(self, n_minibatches: int = 4, generation_config: Optional[Dict] = None, kl_ctl: float = 0.1,
                 adv_norm: bool = True, group_adv_norm: bool = False, group_size: int = 1, discount: float = 1.0,
                 gae_lambda: float = 1.0, eps_clip: float = 0.2, c_clip: Optional[float] = None,
                 max_reward_clip: float = 5.0, reward_output_scaling: float = 1.0, reward_output_bias: float = 0.0,
                 mask_no_eos_with_zero: bool = False, temperature: float = 1.0, use_decoupled_loss: bool = False,
                 behav_imp_weight_cap: Optional[float] = None, recompute_logprob: bool = False,
                 enable_save: bool = True, generation_size: Optional[int] = None, dataset_path: str = "",
                 id2info: Optional[Dict] = None, **_):
        self.gconfig = generation_config if isinstance(generation_config, GenerationHyperparameters) else \
            GenerationHyperparameters(**(generation_config or {}))
        level = "none" if not adv_norm else ("group" if group_adv_norm else "batch")
        self.actor_kw = dict(
            ppo_n_minibatches=n_minibatches, kl_ctl=kl_ctl, group_size=group_size, discount=discount,
            gae_lambda=gae_lambda, eps_clip=eps_clip, c_clip=c_clip, reward_clip=max_reward_clip,
            reward_scaling=reward_output_scaling, reward_bias=reward_output_bias,
            mask_no_eos_with_zero=mask_no_eos_with_zero, temperature=temperature,
            use_decoupled_loss=use_decoupled_loss, behav_imp_weight_cap=behav_imp_weight_cap,
            recompute_logprob=recompute_logprob, max_new_tokens=self.gconfig.max_new_tokens,
            adv_norm=None if level == "none" else AdvNormConfig(mean_level=level, std_level=level, group_size=group_size))
        self.enable_save = enable_save
        # best-of-k: sample `generation_size` answers per prompt and train on the `group_size` best ones
        # (``realhf/impl/model/interface/ppo_interface.py:376-405``); needs the verifier's problem metadata
        self.group_size = group_size
        self.generation_size = generation_size if generation_size is not None else None
        if generation_size is not None:
            assert generation_size >= group_size, (generation_size, group_size)
        self._id2info, self._dataset_path = id2info, dataset_path
        self._actor: Optional[PPOActor] = None
        self._gen_engine = None

    def _algo(self, model: Model) -> PPOActor:
        if self._actor is None:
            self._actor = PPOActor(_actor_config(model, **self.actor_kw), model.module)
        return self._actor

    def save(self, model: Model, save_dir: str):
        if self.enable_save:
            from ...api.io_struct import SaveLoadMeta

            model.module.save(SaveLoadMeta(path=save_dir, weight_format="hf", with_optim=False,
                                           tokenizer=model.tokenizer))

    # ---------------------------------------------------------------------------- generate
    @torch.no_grad()
    def generate(self, model: Model, data: SequenceSample, mb_spec) -> Optional[SequenceSample]:
        """Synchronous PPO: answers come from a rollout engine living in this worker, fed with the
        trainer's current weights by one device copy of the flat parameter buffer; a tensor- / pipeline-parallel
        trainer generates through its own shards and stages instead."""
        from ...impl.backend.native import GenerationEngineModule
        from ...inference.engine import NativeInferenceEngine

        eng = model.module
        eng.onload()
        g = self.gconfig
        hosted = isinstance(eng, GenerationEngineModule)  # a generation-only model: its own engine, its own weights
        model_parallel = not hosted and (eng.grid.tp > 1 or eng.grid.pp > 1)
        if self._gen_engine is None and not model_parallel and not hosted:
            prompt_max = 2048
            srv = InferenceServerConfig(model_path="", dtype=eng.config.dtype, random_seed=1 + eng.grid.global_rank,
                                        context_length=prompt_max + g.max_new_tokens + 64, max_running_requests=256,
                                        disable_cuda_graph=eng.device.type != "cuda")
            self._gen_engine = NativeInferenceEngine(
                srv, model_config=eng.model_config, device=eng.device,
                num_pages=None if eng.device.type == "cuda" else 1024,
                kv_cache_gb=8.0 if eng.device.type == "cuda" else None)
            self._gen_engine.initialize(load_weights=False)
        if not model_parallel and not hosted:
            self._gen_engine.load_weights_from_trainer(eng.model, version=model.version.global_step)
        prompts = [p.data["packed_prompts"].tolist() for p in data.unpack()]
        best_of = self.generation_size is not None and self.generation_size > self.group_size
        n = self.generation_size if best_of else g.n_samples
        stops = list(g.stop_token_ids or [])
        mc = eng.model_config
        for t in (mc.eos_token_id, mc.pad_token_id):
            if t is not None and t not in stops:
                stops.append(int(t))
        if hosted:
            res = eng.generate_tokens(prompts, n, g, stops)
        elif model_parallel:
            # a tensor- / pipeline-parallel trainer generates through its own stages and shards
            # (parallel/pipeline_generate.py; reference: realhf/impl/model/backend/pipe_runner.py:420-646)
            res = eng.generate([p for p in prompts for _ in range(n)], g, stop_token_ids=stops)
        else:
            res = self._gen_engine.generate_blocking(
                [p for p in prompts for _ in range(n)], max_new_tokens=g.max_new_tokens,
                min_new_tokens=g.min_new_tokens, temperature=g.temperature, top_p=g.top_p, top_k=g.top_k,
                greedy=g.greedy, stop_token_ids=stops)
        ids, pmask, lps, no_eos, sl = [], [], [], [], []
        for i, p in enumerate(prompts):
            group = res[i * n:(i + 1) * n]
            if best_of:
                group = self._best_of(model, str(data.ids[i]).split("@")[0], group)
            lens = []
            for r in group:
                seq = p + r.output_tokens
                ids += seq
                pmask += [1] * len(p) + [0] * len(r.output_tokens)
                lps += [0.0] * (len(p) - 1) + list(r.output_logprobs)
                no_eos.append(r.finish_reason != "stop")
                lens.append(len(seq))
            sl.append(lens)
        v = model.version.global_step
        nseq = len(no_eos)
        out = dict(packed_input_ids=torch.tensor(ids, dtype=torch.long), prompt_mask=torch.tensor(pmask, dtype=torch.bool),
                   packed_logprobs=torch.tensor(lps, dtype=torch.float32),
                   seq_no_eos_mask=torch.tensor(no_eos, dtype=torch.bool),
                   version_start=torch.full((nseq,), v, dtype=torch.int), version_end=torch.full((nseq,), v, dtype=torch.int))
        seqlens = dict(packed_input_ids=sl, prompt_mask=sl, packed_logprobs=[[x - 1 for x in g_] for g_ in sl],
                       seq_no_eos_mask=[[1] * len(g_) for g_ in sl], version_start=[[1] * len(g_) for g_ in sl],
                       version_end=[[1] * len(g_) for g_ in sl])
        return SequenceSample(keys=set(out), ids=list(data.ids), seqlens=seqlens, data=out)

    def _best_of(self, model: Model, qid: str, group: List) -> List:
        """The `group_size` answers with the highest (verifier score, generated length), best first."""
        from ...impl.environment.math_code_single_step_env import code_verify_call, extract_code, load_metadata, \
            math_verify_call

        if self._id2info is None:
            assert self._dataset_path, "best-of-k needs dataset_path (or id2info) to score the answers"
            self._id2info = load_metadata(self._dataset_path)[0]
        tok = model.tokenizer
        texts = [tok.decode(r.output_tokens, skip_special_tokens=True) if tok is not None
                 else " ".join(map(str, r.output_tokens)) for r in group]
        if self._id2info.get(qid, {}).get("task", "math") == "code":
            scores = code_verify_call(self._id2info, [extract_code(t) or "" for t in texts], [qid] * len(texts))
        else:
            scores = math_verify_call(self._id2info, texts, [qid] * len(texts))
        order = sorted(range(len(group)), key=lambda j: (float(scores[j]), len(group[j].output_tokens)), reverse=True)
        return [group[j] for j in order[: self.group_size]]

    # --------------------------------------------------------------------------- inference
    @torch.no_grad()
    def inference(self, model: Model, data: SequenceSample, mb_spec) -> Optional[SequenceSample]:
        """Log-probs of the sampled tokens under this model (proximal policy or reference policy)."""
        eng = model.module
        batch = to_padded_batch(data, eng.device)
        logp = self._algo(model).compute_logp(batch, temperature=self.actor_kw["temperature"])
        lens = seqlens_of(data)
        return result_sample(data, dict(logprobs=short_from_padded(logp.float(), lens)), dict(logprobs="short"))

    # -------------------------------------------------------------------------- train_step
    def train_step(self, model: Model, data: SequenceSample, mb_spec) -> List[Dict[str, float]]:
        eng = model.module
        actor = self._algo(model)
        eng.config.mb_spec = mb_spec
        batch = _padded_rl_batch(data, eng.device)
        cfg = actor.config
        if cfg.use_decoupled_loss or cfg.recompute_logprob:
            if "prox_logp" not in batch:
                batch["prox_logp"] = actor.compute_logp(batch, temperature=cfg.temperature)
        vs, ve = batch.pop("version_start", None), batch.pop("version_end", None)
        actor.compute_advantages(batch)
        batch.pop("values", None)
        n_tokens = int(batch["attention_mask"].sum())
        stats = actor.ppo_update(batch)
        cur = model.version.global_step
        model.inc_version()
        extra = dict(n_tokens=float(n_tokens), n_seqs=float(batch["attention_mask"].shape[0]))
        if vs is not None and ve is not None:
            extra.update(head_offpolicyness=float((cur - vs.float()).mean()),
                         tail_offpolicyness=float((cur - ve.float()).mean()),
                         max_head_offpolicyness=float((cur - vs.float()).max()))
        stats[0].update(extra)
        return stats

    def mock(self, type_: str, model: Model, data: SequenceSample) -> SequenceSample:
        return data


class PPOCriticInterface(ModelInterface):
    def __init__(self, n_minibatches: int = 4, kl_ctl: float = 0.1, discount: float = 1.0, gae_lambda: float = 1.0,
                 value_eps_clip: float = 0.2, max_reward_clip: float = 5.0, reward_output_scaling: float = 1.0,
                 reward_output_bias: float = 0.0, mask_no_eos_with_zero: bool = False, group_size: int = 1,
                 value_loss_type: str = "mse", enable_save: bool = True, **_):
        self.critic_kw = dict(ppo_n_minibatches=n_minibatches, value_eps_clip=value_eps_clip,
                              value_loss_type=value_loss_type, mask_no_eos_with_zero=mask_no_eos_with_zero)
        # advantages / returns are derived exactly as the actor does, without normalisation
        self.adv_kw = dict(kl_ctl=kl_ctl, discount=discount, gae_lambda=gae_lambda, reward_clip=max_reward_clip,
                           reward_scaling=reward_output_scaling, reward_bias=reward_output_bias,
                           mask_no_eos_with_zero=mask_no_eos_with_zero, group_size=group_size, adv_norm=None)
        self.enable_save = enable_save
        self._critic: Optional[PPOCritic] = None
        self._adv: Optional[PPOActor] = None

    def _algo(self, model: Model):
        if self._critic is None:
            base = model.module.config
            fields = {f.name for f in dataclasses.fields(PPOCriticConfig)}
            d = {f.name: getattr(base, f.name) for f in dataclasses.fields(type(base)) if f.name in fields}
            d.update(self.critic_kw)
            self._critic = PPOCritic(PPOCriticConfig(**d), model.module)
            self._adv = PPOActor(_actor_config(model, **self.adv_kw), model.module)
        return self._critic

    def save(self, model: Model, save_dir: str):
        if self.enable_save:
            from ...api.io_struct import SaveLoadMeta

            model.module.save(SaveLoadMeta(path=save_dir, weight_format="hf", with_optim=False,
                                           tokenizer=model.tokenizer))

    @torch.no_grad()
    def inference(self, model: Model, data: SequenceSample, mb_spec) -> Optional[SequenceSample]:
        eng = model.module
        batch = to_padded_batch(data, eng.device)
        values = self._algo(model).compute_values(batch).float()
        lens = seqlens_of(data)
        flat = torch.cat([values[b, :n] for b, n in enumerate(lens)])
        return result_sample(data, dict(values=flat), dict(values="token"))

    def train_step(self, model: Model, data: SequenceSample, mb_spec) -> List[Dict[str, float]]:
        eng = model.module
        critic = self._algo(model)
        eng.config.mb_spec = mb_spec
        batch = _padded_rl_batch(data, eng.device)
        assert "values" in batch, "critic training needs the rollout-time values"
        batch.pop("version_start", None), batch.pop("version_end", None)
        batch["values"] = batch["values"].float() * batch["attention_mask"]
        self._adv.compute_advantages(batch)  # writes `returns`, rolls loss_mask to "predict next" alignment
        stats = critic.ppo_update(batch)
        model.inc_version()
        stats[0]["n_tokens"] = float(batch["attention_mask"].sum())
        return stats


register_interface("ppo_actor", PPOActorInterface)
register_interface("ppo_critic", PPOCriticInterface)
