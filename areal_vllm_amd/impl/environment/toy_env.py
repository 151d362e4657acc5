"""Deterministic toy environment for plumbing runs with random-init models: an answer "succeeds" when the
sum of its first tokens plus its length is odd, so about half of all samples succeed and rewards vary
inside a group (non-zero advantages) without any tokenizer or dataset metadata."""
from ...system.agent_api import EnvironmentService, register_environment


class TokenParityEnv(EnvironmentService):
    def __init__(self, n_tokens: int = 16):
        self.n_tokens = n_tokens

    async def step(self, action):
        _, _, output_ids = action[0], action[1], action[2]
        r = [int((sum(o[: self.n_tokens]) + len(o)) % 2) for o in output_ids]
        return None, r, True, False, {}


register_environment("token-parity", TokenParityEnv)
