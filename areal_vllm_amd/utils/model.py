"""Model-family predicates and small module helpers (role of ``areal/utils/model.py:3-25``)."""
from torch import nn

VALID_VISION_MODELS = ["qwen2_vl", "qwen2_5_vl"]
_MOE_TYPES = frozenset({"qwen3_moe"})


def is_qwen2_vl_model(model_type: str) -> bool:
    """Checkpoints whose `model_type` names a Qwen2-VL style vision-language model."""
    return model_type in VALID_VISION_MODELS


def is_qwen3_moe_model(model_type: str) -> bool:
    return model_type in _MOE_TYPES


def disable_dropout_in_model(model: nn.Module) -> None:
    """Set every dropout probability of `model` to zero, in place (RL fine-tuning runs without dropout)."""
    dropouts = (layer for layer in model.modules() if isinstance(layer, nn.Dropout))
    for layer in dropouts:
        layer.p = 0.0
