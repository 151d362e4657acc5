from .client import ArealOpenAI  # noqa: F401
from .types import CompletionWithTokenLogpReward  # noqa: F401
