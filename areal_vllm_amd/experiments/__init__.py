from . import async_ppo_math_exp, ppo_math_exp, sft_exp  # noqa: F401
