from . import math_multi_turn_agent, math_single_step_agent, null_agent  # noqa: F401
