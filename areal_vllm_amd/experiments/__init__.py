from . import async_ppo_math_exp, math_code_eval_exp, null_exp, ppo_math_exp, sft_exp  # noqa: F401
