"""Reusable training loops behind the example entry points."""
