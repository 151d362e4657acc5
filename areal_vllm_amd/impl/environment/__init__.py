from . import math_code_single_step_env, toy_env  # noqa: F401
