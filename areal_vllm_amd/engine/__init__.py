"""Engines: native SPMD trainer, PPO actor / critic, SFT, rollout clients."""
