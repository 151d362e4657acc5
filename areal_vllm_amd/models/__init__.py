"""Native transformer, architecture configs and HuggingFace checkpoint interop."""
from .config import PRESETS, ModelConfig, preset  # noqa: F401
from .transformer import FlatParams, Transformer  # noqa: F401
