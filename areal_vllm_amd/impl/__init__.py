"""Algorithm-side building blocks of the worker runtime: interfaces (PPO actor/critic, SFT, reward),
agents, environments, datasets and numerical utilities."""
