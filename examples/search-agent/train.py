"""Search-agent RL: the policy answers questions by calling `search` / `access` tools over a local corpus,
then states the result in <answer> tags; reward is token F1 against the gold answer.

python -m areal_vllm_amd.launcher.local examples/search-agent/train.py --config examples/search-agent/local_1.5b_example.yaml \
    train_dataset.path=/data/qa_train.jsonl corpus_path=/data/corpus.jsonl actor.path=/ckpt/Qwen2.5-1.5B

Training rows: {"question": ..., "answer": ... | [...]}; corpus rows: {"url": ..., "title": ..., "text": ...}.
"""
import json
import sys
from dataclasses import dataclass

from areal_vllm_amd.api.cli_args import GRPOConfig
from areal_vllm_amd.dataset import register_dataset
from areal_vllm_amd.dataset.synthetic import ListDataset
from areal_vllm_amd.impl.environment.local_search_env import BM25Index, LocalSearchEnv, f1_answer_reward
from areal_vllm_amd.trainer.rl import run_grpo
from areal_vllm_amd.workflow.tool_agent import ToolAgentWorkflow

SYSTEM = ("Answer the question. You may call the tools to look things up; when you know the answer, reply with "
          "<answer> ... </answer>.")


@dataclass
class SearchAgentConfig(GRPOConfig):
    max_turns: int = 32
    corpus_path: str = ""
    tool_call_parser: str = "qwen25"
    max_observation_tokens: int = 512


def qa_dataset(path, split, rank, world_size, **kw):
    with open(path) as f:
        rows = [json.loads(l) for l in f if l.strip()]
    out = [dict(r, query_id=str(r.get("id", i)), messages=[{"role": "system", "content": SYSTEM},
                                                          {"role": "user", "content": r["question"]}])
           for i, r in enumerate(rows)]
    return ListDataset(out[rank::world_size])


def build(config, tokenizer, reward_fn, log_dir):
    with open(config.corpus_path) as f:
        index = BM25Index([json.loads(l) for l in f if l.strip()])
    mk = lambda g, scope: ToolAgentWorkflow(  # noqa: E731
        reward_fn, g, tokenizer, env_factory=lambda: LocalSearchEnv(index), max_turns=config.max_turns,
        tool_call_parser=config.tool_call_parser, max_observation_tokens=config.max_observation_tokens,
        max_total_tokens=config.sglang.context_length - 64, rollout_stat_scope=scope,
        agent_stats_keys=config.actor.log_agent_stats_keys)
    return mk(config.gconfig, "rollout"), mk(config.gconfig.new(temperature=0.6), "eval-rollout")


def main(args):
    register_dataset(".jsonl", qa_dataset)
    run_grpo(args, reward_fn=f1_answer_reward, workflow_builder=build, config_cls=SearchAgentConfig)


if __name__ == "__main__":
    main(sys.argv[1:])
