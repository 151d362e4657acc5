"""Model-family predicates (``areal/utils/model.py:3-25``)."""
import torch

VALID_VISION_MODELS = ["qwen2_vl", "qwen2_5_vl"]


def is_qwen2_vl_model(model_type: str) -> bool:
    return model_type in ("qwen2_vl", "qwen2_5_vl")


def is_qwen3_moe_model(model_type: str) -> bool:
    return model_type == "qwen3_moe"


def disable_dropout_in_model(model: torch.nn.Module) -> None:
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0
