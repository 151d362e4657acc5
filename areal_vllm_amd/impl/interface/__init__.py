from . import fused_interface, math_rw_interface, ppo_interface, sft_interface  # noqa: F401
