"""Rollout workflows: single-turn RLVR, multi-turn with retry, vision RLVR."""
