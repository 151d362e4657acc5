"""Convert a config of another system into a YAML for this framework.

    python examples/config_converter.py --convert_src AReaL    --src_config_path legacy.yaml \
        --template_path examples/math/gsm8k_grpo.yaml --output_path out.yaml
    python examples/config_converter.py --convert_src OpenRLHF --src_script_path train.sh \
        --template_path examples/math/gsm8k_grpo.yaml --output_path out.yaml

Sources: the worker-runtime ("legacy") experiment YAML / quickstart options, or an OpenRLHF launch script
(its ``--flag value`` arguments are read out of the shell text). The template supplies every key the
source does not mention. Keys without a counterpart are reported, never silently dropped. Role of the
reference's ``examples/config_converter.py``; the mapping is written as dotted-path rules instead of one
table entry per key.
"""
import argparse
import os
import re
import shlex
import sys
from typing import Any, Callable, Dict, List, Optional, Tuple

import yaml

Rule = Tuple[str, Optional[str], Optional[Callable[[Any], Any]]]  # (source pattern, target path | None, map)

_bf16 = lambda x: "bfloat16" if x in (True, "true", "True", None) else "float16"  # noqa: E731

# `*` matches one dotted component and is substituted into the target.
LEGACY_RULES: List[Rule] = [
    ("experiment_name", "experiment_name", None), ("trial_name", "trial_name", None), ("seed", "seed", None),
    ("allocation_mode", "allocation_mode", lambda s: re.sub(r"m(\d)", r"t\1", str(s)).replace("sglang", "native")
        .replace("vllm", "native")),
    ("n_nodes", "cluster.n_nodes", None), ("n_gpus_per_node", "cluster.n_gpus_per_node", None),
    ("cluster.fileroot", "cluster.fileroot", None), ("cluster.name_resolve.*", "cluster.name_resolve.*", None),
    ("cluster.mount", "launcher.slurm.mount", None),
    ("cpus_per_model_worker", "launcher.trainer_cpus_per_gpu", None),
    ("mem_per_model_worker", "launcher.trainer_mem_per_gpu", None),
    ("cpus_per_generation_server", "launcher.inference_server_cpus_per_gpu", None),
    ("mem_per_generation_server", "launcher.inference_server_mem_per_gpu", None),
    ("actor.path", "actor.path", None), ("actor.init_from_scratch", "actor.init_from_scratch", None),
    ("actor.gradient_checkpointing", "actor.gradient_checkpointing", None), ("actor.bf16", "actor.dtype", _bf16),
    ("actor.optimizer.*", "actor.optimizer.*", None),
    ("actor.sglang.*", "sglang.*", None), ("actor.vllm.*", None, None),
    ("ref.path", "ref.path", None), ("ref.bf16", "ref.dtype", _bf16),
    ("critic.path", "critic.path", None), ("critic.optimizer.*", "critic.optimizer.*", None),
    ("dataset.path", "train_dataset.path", None), ("dataset.max_prompt_len", "train_dataset.max_length", None),
    ("dataset.train_bs_n_seqs", "train_dataset.batch_size", None),
    ("group_size", "gconfig.n_samples", None),
    ("ppo.gen.max_new_tokens", "gconfig.max_new_tokens", None), ("ppo.gen.min_new_tokens", "gconfig.min_new_tokens", None),
    ("ppo.gen.temperature", "gconfig.temperature", None), ("ppo.gen.top_p", "gconfig.top_p", None),
    ("ppo.gen.top_k", "gconfig.top_k", None), ("ppo.gen.greedy", "gconfig.greedy", None),
    ("ppo.ppo_n_minibatches", "actor.ppo_n_minibatches", None), ("ppo.eps_clip", "actor.eps_clip", None),
    ("ppo.c_clip", "actor.c_clip", None), ("ppo.kl_ctl", "actor.kl_ctl", None),
    ("ppo.discount", "actor.discount", None), ("ppo.gae_lambda", "actor.gae_lambda", None),
    ("ppo.reward_output_scaling", "actor.reward_scaling", None), ("ppo.reward_output_bias", "actor.reward_bias", None),
    ("ppo.max_reward_clip", "actor.reward_clip", None), ("ppo.recompute_logprob", "actor.recompute_logprob", None),
    ("ppo.use_decoupled_loss", "actor.use_decoupled_loss", None),
    ("ppo.behav_imp_weight_cap", "actor.behav_imp_weight_cap", None),
    ("ppo.disable_value", None, None), ("ppo.value_eps_clip", "critic.eps_clip", None),
    ("max_head_offpolicyness", "rollout.max_head_offpolicyness", None),
    ("max_concurrent_rollouts", "rollout.max_concurrent_rollouts", None),
    ("new_tokens_per_chunk", None, None), ("flush_request_timeout", "rollout.request_timeout", None),
    ("actor_train.mb_spec.*", "actor.mb_spec.*", None), ("ref_inf.mb_spec.*", "ref.mb_spec.*", None),
    ("exp_ctrl.total_train_epochs", "total_train_epochs", None), ("exp_ctrl.save_freq_steps", "saver.freq_steps", None),
    ("exp_ctrl.save_freq_epochs", "saver.freq_epochs", None), ("exp_ctrl.save_freq_secs", "saver.freq_secs", None),
    ("exp_ctrl.ckpt_freq_steps", "recover.freq_steps", None), ("exp_ctrl.ckpt_freq_secs", "recover.freq_secs", None),
    ("exp_ctrl.eval_freq_steps", "evaluator.freq_steps", None), ("exp_ctrl.eval_freq_epochs", "evaluator.freq_epochs", None),
    ("recover_mode", "recover.mode", None), ("recover_retries", "recover.retries", None),
    ("wandb.*", "stats_logger.wandb.*", None), ("swanlab.*", "stats_logger.swanlab.*", None),
    ("tensorboard.*", "stats_logger.tensorboard.*", None),
]

OPENRLHF_RULES: List[Rule] = [
    ("pretrain", "actor.path", None), ("actor_num_nodes", "cluster.n_nodes", None),
    ("actor_num_gpus_per_node", "cluster.n_gpus_per_node", None), ("async_train", "async_training", lambda x: True),
    ("ckpt_path", "cluster.fileroot", os.path.dirname), ("gradient_checkpointing", "actor.gradient_checkpointing", lambda x: True),
    ("bf16", "actor.dtype", _bf16), ("vllm_gpu_memory_utilization", "sglang.mem_fraction_static", float),
    ("actor_learning_rate", "actor.optimizer.lr", float), ("critic_learning_rate", "critic.optimizer.lr", float),
    ("l2", "actor.optimizer.weight_decay", float), ("max_norm", "actor.optimizer.gradient_clipping", float),
    ("lr_warmup_ratio", "actor.optimizer.warmup_steps_proportion", float), ("lr_scheduler", "actor.optimizer.lr_scheduler_type", str),
    ("adam_betas", None, None), ("seed", "seed", int), ("max_epochs", "total_train_epochs", int),
    ("num_episodes", None, None), ("rollout_batch_size", "train_dataset.batch_size", int),
    ("train_batch_size", None, None), ("micro_train_batch_size", None, None),
    ("n_samples_per_prompt", "gconfig.n_samples", int), ("generate_max_len", "gconfig.max_new_tokens", int),
    ("prompt_max_len", "train_dataset.max_length", int), ("temperature", "gconfig.temperature", float),
    ("top_p", "gconfig.top_p", float), ("eps_clip", "actor.eps_clip", float), ("value_clip", "critic.eps_clip", float),
    ("init_kl_coef", "actor.kl_ctl", float), ("gamma", "actor.discount", float), ("lambd", "actor.gae_lambda", float),
    ("prompt_data", "train_dataset.path", str), ("save_steps", "saver.freq_steps", int),
    ("eval_steps", "evaluator.freq_steps", int), ("use_wandb", "stats_logger.wandb.mode", lambda x: "online"),
    ("wandb_project", "stats_logger.wandb.project", str), ("wandb_run_name", "stats_logger.wandb.name", str),
]


def flatten(d: Dict[str, Any], prefix: str = "") -> Dict[str, Any]:
    out = {}
    for k, v in d.items():
        key = f"{prefix}.{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten(v, key))
        else:
            out[key] = v
    return out


def set_path(d: Dict[str, Any], path: str, value: Any):
    parts = path.split(".")
    for p in parts[:-1]:
        if not isinstance(d.get(p), dict):
            d[p] = {}
        d = d[p]
    d[parts[-1]] = value


def match_rule(key: str, rules: List[Rule]):
    """-> (target path or None, map fn) for the first rule matching `key`; (False, None) if none does."""
    for pat, target, fn in rules:
        if "*" in pat:
            head = pat[:-1]
            if key.startswith(head):
                rest = key[len(head):]
                return (target[:-1] + rest if target else None), fn
        elif pat == key:
            return target, fn
    return False, None


def apply_rules(src: Dict[str, Any], template: Dict[str, Any], rules: List[Rule]):
    out = yaml.safe_load(yaml.safe_dump(template))  # deep copy that keeps ${...} strings
    unmapped, dropped = [], []
    for key, value in src.items():
        target, fn = match_rule(key, rules)
        if target is False:
            unmapped.append(key)
        elif target is None:
            dropped.append(key)
        else:
            set_path(out, target, fn(value) if fn is not None else value)
    return out, unmapped, dropped


def parse_openrlhf_script(text: str) -> Dict[str, Any]:
    """Flags of the (last) python / ray-job command in a launch script: `--a b` -> {a: b}, `--flag` -> True."""
    text = text.replace("\\\n", " ")
    toks = shlex.split(text, comments=True)
    args: Dict[str, Any] = {}
    i = 0
    while i < len(toks):
        t = toks[i]
        if t.startswith("--") and len(t) > 2:
            name = t[2:].replace("-", "_")
            if "=" in name:
                name, _, val = name.partition("=")
                args[name] = val
            elif i + 1 < len(toks) and not toks[i + 1].startswith("--"):
                args[name] = toks[i + 1]
                i += 1
            else:
                args[name] = True
        i += 1
    if "vllm_num_engines" in args:
        d, t = int(args.pop("vllm_num_engines")), int(args.pop("vllm_tensor_parallel_size", 1))
        n = int(args.get("actor_num_nodes", 1)) * int(args.get("actor_num_gpus_per_node", 8))
        args["__alloc"] = f"native:d{d}t{t}+native:d{n}"
    return args


def convert(src_kind: str, src_path: str, template_path: str):
    with open(template_path) as f:
        template = yaml.safe_load(f)
    if src_kind == "AReaL":
        with open(src_path) as f:
            src = flatten(yaml.safe_load(f) or {})
        out, unmapped, dropped = apply_rules(src, template, LEGACY_RULES)
    else:
        with open(src_path) as f:
            src = parse_openrlhf_script(f.read())
        alloc = src.pop("__alloc", None)
        out, unmapped, dropped = apply_rules(src, template, OPENRLHF_RULES)
        if alloc:
            out["allocation_mode"] = alloc
    return out, unmapped, dropped


def main(argv=None):
    ap = argparse.ArgumentParser(description="Convert a legacy / OpenRLHF config into this framework's YAML")
    ap.add_argument("--convert_src", choices=["OpenRLHF", "AReaL"], default="AReaL")
    ap.add_argument("--src_script_path")
    ap.add_argument("--src_config_path")
    ap.add_argument("--template_path", required=True)
    ap.add_argument("--output_path", default="output.yaml")
    a = ap.parse_args(argv)
    src = a.src_config_path if a.convert_src == "AReaL" else a.src_script_path
    if not src:
        ap.error("give --src_config_path (AReaL) or --src_script_path (OpenRLHF)")
    out, unmapped, dropped = convert(a.convert_src, src, a.template_path)
    with open(a.output_path, "w") as f:
        yaml.safe_dump(out, f, sort_keys=False, allow_unicode=True)
    for k in dropped:
        print(f"[no counterpart] {k}", file=sys.stderr)
    for k in unmapped:
        print(f"[not converted ] {k}", file=sys.stderr)
    print(f"wrote {a.output_path} ({len(unmapped)} keys not converted, {len(dropped)} without counterpart)")


if __name__ == "__main__":
    main()
